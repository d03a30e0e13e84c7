// Householder QR (placeholder until the kernels land; fails loudly, never falls back to the CPU).
#include "common.h"
namespace fh {
template <typename T> long geqrf_dev(MatV<T>, MatV<T>, idx_t) { die("qr: not implemented yet", __FILE__, __LINE__); }
template <typename T> void apply_householder_sequence_left_dev(MatV<const T>, MatV<const T>, MatV<T>, bool)
{
	die("apply_householder: not implemented yet", __FILE__, __LINE__);
}
template long geqrf_dev<double>(MatV<double>, MatV<double>, idx_t);
template long geqrf_dev<float>(MatV<float>, MatV<float>, idx_t);
template void apply_householder_sequence_left_dev<double>(MatV<const double>, MatV<const double>, MatV<double>, bool);
template void apply_householder_sequence_left_dev<float>(MatV<const float>, MatV<const float>, MatV<float>, bool);
} // namespace fh
