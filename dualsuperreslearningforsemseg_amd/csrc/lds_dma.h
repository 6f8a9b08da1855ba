// LDS-DMA helpers (gfx950): buffer_load_dwordx4 ... lds, counted vmcnt waits, bare s_barrier.  Shared by conv_planes.hip and convt_dma.hip.
#pragma once
#include "common.h"

namespace dsrl {

using dma_u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// One LDS-DMA piece: 64 lanes x 16 bytes, global address = descriptor base + voff (bounds-checked) + soff (unchecked), LDS address = lds + 16 * lane.
// Issued from inline asm: hipcc tracks a builtin LDS-DMA as an LDS store and makes every later ds_read wait for it (vmcnt(0) in front of the fragment
// reads of the OTHER slot); the waits are placed by hand instead (s_waitcnt_vm below).
// M0 is written inside the asm without being declared: hipcc rejects "m0" as a clobber ("reserved register ... undefined behaviour"), and nothing the
// compiler generates for the translation units that include this header uses M0 (no movrel indexing, builtin LDS-DMA, sendmsg or GWS) -
// tests/test_abi_and_host.py disassembles their objects and fails the build check if any instruction other than these s_mov_b32 ever reads or writes m0.
__device__ __forceinline__ void lds_dma16(dma_u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// raw buffer descriptor (stride 0): {base[31:0], base[47:32], bytes, flags} - what __builtin_amdgcn_make_buffer_rsrc builds, as four SGPRs for the asm above
__device__ __forceinline__ dma_u32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long b = (unsigned long long)p;
    dma_u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b); r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes); r.w = 0x00020000u;
    return r;
}
template <int N> __device__ __forceinline__ void s_waitcnt_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace dsrl
