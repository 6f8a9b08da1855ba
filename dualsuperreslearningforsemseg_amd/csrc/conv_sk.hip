// 128x128 block tiles with two K groups (8 waves, wave tile 64x64) and split-K ACROSS workgroups reduced inside the launch (round 5).
//
// Why: the M = 4096 layers of the step (layer3's 69 convs, layer4, the ASPP 1x1s) have 64..256 tiles of 128x128, and the 64x64 / four-K-group plan
// of rounds 2-4 stages (64 + 64) rows x 4 bytes per channel and CU - 1.18 MB per CU on the layer3 3x3 conv, 302 MB over the chip for 10 MB of
// operands - at 0.67 KB of LDS fragment reads per MFMA.  Here a CU stages (128 + 128) rows for a QUARTER of the K loop (half the bytes per CU), a wave
// owns 2 x 2 tiles of 32x32 (0.33 KB of fragment reads per MFMA), and the partial tiles of the `splits` workgroups that share an output tile meet in
// the launch itself: conv_igemm_split_kernel<..., COOP = true> (conv_split_kernel.h) - write-through partials, one ticket per tile, the last arriver
// sums in a fixed order and runs the ordinary epilogue.  Same arithmetic (f16x3 / f16x1, pre-split or on-the-fly filters) as every other build of
// that template; results equal slabs + splitk_reduce_kernel bit for bit.
#include "common.h"
#include "conv_common.h"
#include "conv_split_kernel.h"
#include <algorithm>

namespace dsrl {

constexpr size_t kSkStages = (size_t)2 * (128 + 128) * 2 * 64 * 2;                 // two full-step stages of 256 rows x 2 planes x 64 B per K group, two groups
constexpr size_t kSkReduce = (size_t)2 * 2 * 2 * 4 * 256 * 16 + 8192;              // the K-group reduction (every group's accumulators) + the BatchNorm-sum exchange
constexpr size_t kSkLds = kSkStages > kSkReduce ? kSkStages : kSkReduce;
static_assert(kSkLds <= 160 * 1024, "LDS");
size_t sk_lds_bytes() { return kSkLds; }

bool sk_supported(int cfg, int kg, int npl, bool f16, bool w_split) {
    (void)w_split;
    return cfg == 0 && (kg == 2 || kg == 1) && f16 && (npl == 1 || npl == 2);
}

template <bool DGRAD, int NPL, int ARITH, bool STR1, int KG>
static int launch_one_kg(const ConvArgs& a, hipStream_t st) {
    auto* k = conv_igemm_split_kernel<2, 2, 2, 2, DGRAD, NPL, KG, ARITH, STR1, true>;
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSkLds);
    (void)attr;
    const dim3 grid((unsigned)(a.mtiles * a.ntiles), 1u, (unsigned)a.splits);
    // one K group: the LDS of the ordinary 128x128 launch (two stages of 256 rows x NPL planes x 64 B, or the tile parked for the fast BatchNorm-sum epilogue)
    const size_t lds = KG > 1 ? kSkLds : std::max((size_t)2 * (128 + 128) * NPL * 64, a.bn_fast ? (size_t)128 * 128 * 4 : (size_t)0);
    hipLaunchKernelGGL(k, grid, dim3(256 * KG), lds, st, a);
    return launch_status("conv_igemm_split_kernel<128x128, cooperative split-K>");
}
// one K group (DSRL_SK_COOP1, round 5): the planner's ordinary 128x128 / split-K plan with the reduction moved into the launch - no slabs + reduce launch
template <bool DGRAD, int NPL, int ARITH, bool STR1>
static int launch_one(const ConvArgs& a, hipStream_t st) {
    return a.kg > 1 ? launch_one_kg<DGRAD, NPL, ARITH, STR1, 2>(a, st) : launch_one_kg<DGRAD, NPL, ARITH, STR1, 1>(a, st);
}

int launch_sk_igemm(const ConvArgs& a, bool dgrad, bool str1, int npl, hipStream_t st) {
    if (a.splits > 1 && a.tickets != nullptr) {
        if (a.coop_slab == nullptr || a.mtiles * a.ntiles > kCoopMaxTiles) { set_error("cooperative split-K: no slab, or more than %d tiles", kCoopMaxTiles); return DSRL_E_BADARG; }
        if ((long long)a.splits * a.mtiles * a.ntiles * 128 * 128 * 4 >= (1ll << 31)) { set_error("cooperative split-K: partial tiles exceed the 2 GiB descriptor range"); return DSRL_E_UNSUPPORTED; }
    }
    if (npl == 2) {
        if (a.w_split) {
            if (dgrad) return str1 ? launch_one<true, 2, 2, true>(a, st) : launch_one<true, 2, 2, false>(a, st);
            return launch_one<false, 2, 2, false>(a, st);
        }
        return dgrad ? launch_one<true, 2, 1, false>(a, st) : launch_one<false, 2, 1, false>(a, st);
    }
    if (a.w_split) return dgrad ? launch_one<true, 1, 2, false>(a, st) : launch_one<false, 1, 2, false>(a, st);
    return dgrad ? launch_one<true, 1, 1, false>(a, st) : launch_one<false, 1, 1, false>(a, st);
}

}  // namespace dsrl
