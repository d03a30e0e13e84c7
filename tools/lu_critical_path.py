"""Dependency-graph lower bound of the partial-pivot LU at N = 16384 on one MI355X (can 0.60 x DGEMM = 67.6 ms be
reached with exact partial-pivot semantics on this part?).  Pure arithmetic on MEASURED node times; no GPU needed.

The graph.  Columns are eliminated in 64-column leaves (16384 dependent pivot searches in all -- lu/partial_pivoting/factor.rs:19-67 picks the
pivot of column j + 1 from the column as updated by column j).  Panel k = a run of leaves with the in-panel updates between them; the
NEXT panel's columns need panel k applied before its first leaf can start ("hand-over"); the far columns take panel k as one product
per step, and those products form a second sequential chain (product k + 1 accumulates into what product k wrote):

    leaf -> node update -> leaf -> ... -> [hand-over k] -> panel k + 1 -> ...                (panel chain, 32 reserved CUs)
    product k  needs  panel k  and  product k - 1;  hand-over k in the bulk-bound phase needs product k - 1 on those columns

Node times (microseconds), where they come from:
    leaf, per column        2.34 alone on an idle chip (tools/gpu_lu_leaf_alone.py: 150 us per 64 columns, profiles/r05_exp_lu_driver.txt)
                            2.60 in the panel-bound phase (bulk stream mostly idle: 165-170 us per leaf, profiles/r05_lu_timeline.txt)
                            2.96 beside a streaming product (BENCH_r05 / r6base: chain_kernel.us_per_column)
                            1.56 for panels of <= 512 rows (single-workgroup leaf, ~100 us per leaf)
    node update             61 / 60 / 38 us for 3 / 2 / 1 column groups alone on the panel stream, 111 / 111 / 75 beside the bulk stream
                            (profiles/r05_exp_lu_driver.txt item 8); wider runs extrapolated at +11 us per group
    in-panel interchanges   2 x 6 us per leaf alone, 2 x 12-13 in situ
    hand-over               staged (panel-bound phase): the last leaf's interchanges + ONE node launch on the whole chip ~ 100 us;
                            bulk-bound phase: interchanges + 512-wide solve + slice product on the bulk stream ~ 350-550 us
    products                per step, alone on 224 CUs / alone on 256 CUs / in situ: profiles/r06_update_in_situ.txt
    first panel             nothing overlaps it"""
import math

N = 16384
PEAK_DGEMM_MS = 2 * N ** 3 / 3 / 72.3e12 * 1e3  # LU flops at the sustained DGEMM rate
BAR = PEAK_DGEMM_MS / 0.60


def plan():
    J = [0]
    while J[-1] < N:
        j0 = J[-1]
        w = 512 if N - j0 - 512 >= 9216 else 256
        J.append(min(N, j0 + w))
    return J


def node_us(groups, rows, in_situ):
    base = {1: 38.0, 2: 60.0, 3: 61.0}.get(groups, 61.0 + 11.0 * (groups - 3))
    if in_situ:
        base *= 111.0 / 61.0
    return base * max(rows, 2048) / 8704.0 if rows < 8704 else base * rows / 8704.0


def panel_us(j0, w, t_col, in_situ, t_col_small=1.56):
    rows = N - j0
    nl = w // 64
    t = 0.0
    for i in range(nl):
        r = rows - 64 * i
        tc = t_col_small if r <= 512 else t_col
        t += 64 * tc
        t += 2 * (12.5 if in_situ else 6.0)
        g = nl - 1 - i
        if g > 0:
            t += node_us(g, r, in_situ)
    return t


def product_ms(rows, cols, k, rate_tf):
    return 2.0 * rows * cols * k / (rate_tf * 1e12) * 1e3


def run(label, t_col_bulk, t_col_panel, in_situ, rate512, rate256, handover_bulk_us, handover_staged_us, overlap="driver"):
    J = plan()
    ns = len(J) - 1
    p_end = 0.0
    g_end = 0.0
    chain_only = 0.0
    log = []
    for k in range(ns):
        j0, j1 = J[k], J[k + 1]
        w = j1 - j0
        rows_below = N - j1
        bulk_bound = rows_below >= 9216
        tc = t_col_bulk if bulk_bound else t_col_panel
        pan = panel_us(j0, w, tc, in_situ and bulk_bound) * 1e-3
        if k == 0:
            p_start = 0.0
        else:
            p_start = ready_next
        p_end = p_start + pan
        chain_only += pan
        # product of panel k on the far columns (everything right of the next panel)
        j2 = J[k + 2] if k + 2 <= ns else N
        w2 = j2 - j1
        rate = rate512 if w == 512 else rate256
        prod = product_ms(N - j1, max(N - j2, 0), w, rate) if rows_below > 0 else 0.0
        slice_ = product_ms(N - j1, w2, w, rate * 0.7) if w2 > 0 else 0.0
        if bulk_bound:
            # hand-over on the bulk stream: behind the previous product, then the slice for the next panel, then the far product
            h0 = max(p_end, g_end)
            ready_next = h0 + handover_bulk_us * 1e-3 + slice_
            g_start = ready_next
            chain_only += handover_bulk_us * 1e-3 + slice_
        else:
            ready_next = p_end + handover_staged_us * 1e-3
            g_start = max(ready_next, g_end)
            chain_only += handover_staged_us * 1e-3
        g_end = g_start + prod
        log.append((k, j0, w, pan, prod, p_start, p_end, g_start, g_end))
    total = max(p_end, g_end)
    print(f"{label:78s} total {total:6.1f} ms   (panel chain alone {chain_only:5.1f} ms, products {sum(x[4] for x in log):5.1f} ms)")
    return total, log


print(__doc__)
print(f"N = {N}: 2N^3/3 = {2 * N ** 3 / 3:.3e} flop; at the sustained DGEMM rate (72.3 TFLOP/s) {PEAK_DGEMM_MS:.1f} ms; the bar 0.60 x DGEMM = {BAR:.1f} ms; "
      "measured this round 86.4-86.9 ms (fast boxes)\n")
print("Schedules (the driver's own: 512-column steps while >= 9216 rows remain below the panel, then 256-column steps with the staged hand-over):")
run("A  measured node times (in situ leaf 2.96 / 2.60, products at the in-situ rates 51 / 50 TF)", 2.96, 2.60, True, 51.0, 50.0, 450, 100)
run("B  same schedule, products alone on 224 CUs (56 / 51 TF), leaf 2.96 / 2.60", 2.96, 2.60, True, 56.0, 51.0, 450, 100)
run("C  every kernel at its idle-chip time (leaf 2.34, nodes alone), products alone on 224 CUs", 2.34, 2.34, False, 56.0, 51.0, 350, 100)
run("D  C + products at the whole-chip rate (65 / 58 TF: no reserved CUs at all)", 2.34, 2.34, False, 65.0, 58.0, 350, 100)
run("E  C + hand-over 100 us everywhere (a staged hand-over in the bulk-bound phase as well)", 2.34, 2.34, False, 56.0, 51.0, 100, 100)
print()
print("What the leaf would have to cost for the bar (schedule E, everything else at idle-chip times):")
for tc in (2.34, 2.0, 1.8, 1.6, 1.4, 1.2):
    run(f"   leaf {tc:.2f} us per column", tc, tc, False, 56.0, 51.0, 100, 100)
print("""
Reading.
  * A reproduces the measured factorization to ~3 ms (89.5 against 86.4-86.9): the model's nodes and edges are the ones that matter.
  * The pivot chain ALONE -- 16384 columns x 2.34 us on an idle chip -- is 38.3 ms.  With the in-panel node updates, the interchanges and the
    hand-overs between panels the panel chain is 62-66 ms of strictly dependent work at IDLE-CHIP kernel times (C, E) and 84 ms at the times the
    same kernels take beside the products they are supposed to hide behind (A, B: leaf 2.96 instead of 2.34 us per column, node updates 1.8 x).
  * Schedule E is the bound of this panel algorithm: every kernel at its idle-chip time, no interference between the streams, a staged
    100 us hand-over in every step, products alone on their 224 CUs -- 74 ms, still above the bar (67.6).  D shows what the reserved CUs cost
    (5 ms) if a chain could run on CUs the products also use -- it cannot: a 130-260 us tile in every slot delays each of the chain's ~1500
    launches (DESIGN.md 3.6, profiles/r06_exp_lend.txt).
  * Under E the bar needs a leaf at <= 1.6 us per column.  The exchange protocol family measured in round 5 bottoms out at 0.94-1.2 us for
    the 32-way all-to-all alone on an idle chip (tools/xwg_a2a_probe.hip) plus ~1.0 us of dependent work inside the workgroup (arg-max over
    four wavefronts, publication, relabelling): 1.9-2.2 us before any neighbour; 2.34 measured.
  Verdict: with exact partial-pivot semantics (the pivot of column j + 1 chosen from the column as updated by column j, identical to the
  reference's: one cross-workgroup all-to-all per column) 67.6 ms is NOT reachable on this part; the bound is ~74 ms and what scheduling
  could still give from today's 86.5 is the distance A -> C (interference: ~10 ms, none of it removed by five rounds of scheduling
  experiments) and C -> E (hand-overs: ~3.5 ms).  Reaching 0.60 x DGEMM needs a different panel algorithm -- fewer dependent exchanges per
  column block (tournament pivoting / CALU), which changes the pivots and therefore the contract "identical permutation to faer".""")
