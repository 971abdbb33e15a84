// The generated quotient kernels (quotient_kernels.inc, written by tools/gen_quotient_kernels.py) and their registry --
// a translation unit of their own: the straight-line kernels take minutes to compile.
#include "quotient29.hip.hpp"

using namespace shp;

#include "quotient_kernels.inc"

namespace shp {

QuotientKernelFn quotient_kernel_for(uint64_t hash) {
    for (const QuotientKernelEntry& q : QUOTIENT_KERNELS)
        if (q.hash == hash) return q.fn;
    return nullptr;
}

}  // namespace shp
