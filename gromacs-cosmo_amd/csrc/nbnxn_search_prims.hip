// Device-wide scan and stable radix sorts used by the pair search, from hipCUB (rocPRIM).
// Kept in a translation unit of their own: see nbnxn_search.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "nbnxn_search.h"

namespace {

hipError_t reserve(nbx_prim_tmp &tmp, size_t need)
{
    if (need <= tmp.bytes) { return hipSuccess; }
    if (tmp.p) { hipError_t e = hipFree(tmp.p); tmp.p = nullptr; tmp.bytes = 0; if (e != hipSuccess) { return e; } }
    const size_t n = need + need / 4 + 256;
    hipError_t   e = hipMalloc(&tmp.p, n);
    if (e == hipSuccess) { tmp.bytes = n; }
    return e;
}

} // namespace

hipError_t nbx_prim_exclusive_sum(nbx_prim_tmp &tmp, const int *in, int *out, int n, hipStream_t st)
{
    size_t     need = 0;
    hipError_t e    = hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, n, st);
    if (e != hipSuccess) { return e; }
    e = reserve(tmp, need);
    if (e != hipSuccess) { return e; }
    size_t have = tmp.bytes;
    return hipcub::DeviceScan::ExclusiveSum(tmp.p, have, in, out, n, st);
}

hipError_t nbx_prim_sort_u64_int(nbx_prim_tmp &tmp, const unsigned long long *k_in, unsigned long long *k_out, const int *v_in,
                                 int *v_out, int n, int end_bit, hipStream_t st)
{
    size_t     need = 0;
    hipError_t e    = hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, n, 0, end_bit, st);
    if (e != hipSuccess) { return e; }
    e = reserve(tmp, need);
    if (e != hipSuccess) { return e; }
    size_t have = tmp.bytes;
    return hipcub::DeviceRadixSort::SortPairs(tmp.p, have, k_in, k_out, v_in, v_out, n, 0, end_bit, st);
}

hipError_t nbx_prim_sort_u32_sci(nbx_prim_tmp &tmp, const unsigned *k_in, unsigned *k_out, const nbx_sci_t *v_in, nbx_sci_t *v_out,
                                 int n, int end_bit, hipStream_t st)
{
    size_t     need = 0;
    hipError_t e    = hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, n, 0, end_bit, st);
    if (e != hipSuccess) { return e; }
    e = reserve(tmp, need);
    if (e != hipSuccess) { return e; }
    size_t have = tmp.bytes;
    return hipcub::DeviceRadixSort::SortPairs(tmp.p, have, k_in, k_out, v_in, v_out, n, 0, end_bit, st);
}

void nbx_prim_free(nbx_prim_tmp &tmp)
{
    if (tmp.p) { (void)hipFree(tmp.p); }
    tmp.p = nullptr; tmp.bytes = 0;
}
