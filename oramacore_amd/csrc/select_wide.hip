// select_wide.hip — the two selection kernels of the second half of round 5 in a code object of their own:
//   select_tiny_kernel        <= 512 values in one workgroup, ranks by counting, no sort
//   pairs_reduce_wide_kernel  a part of <= 32 768 values in registers, one bound, one cut
// Their source — and the device helpers they share with the other selection kernels — is select.hip, compiled here as its
// second unit; select.hip says why they do not live in the unit "select".
#define ORAMA_SELECT_UNIT 2
#pragma clang diagnostic ignored "-Wunused-function"
#pragma clang diagnostic ignored "-Wunused-const-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "select.hip"
