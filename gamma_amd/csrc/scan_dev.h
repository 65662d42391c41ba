// scan_dev.h -- device helpers shared by the list-scan kernels (kernels.hip: k_ivfpq_scan_pair, scan_lm.hip:
// k_scan_lm): distance keys, LUT stores with ds_write_addtid_b32, SDWA gather addresses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "device_math.h"

namespace gh {

// order-preserving key of a distance in "smaller is better" form (the selection kernels' key)
template <bool L2>
__device__ __forceinline__ uint32_t dis_key(float v) {
    const uint32_t k = f2key(v);
    return L2 ? k : ~k;
}
constexpr uint32_t KEY_SENTINEL = 0xff800000u;   // key of the filtered-entry marker (+inf / -inf)

// LDS byte address of LUT entry (m, code byte k of w): (byte << 2) + 1024 * m with the LUT at LDS address 0.
// One SDWA shift selects the byte and scales it (hipcc emits an extract and a shift-add: two VALU ops per
// look-up, a third of the scan loop's VALU work); the row offset rides in the ds_read's immediate.
__device__ __forceinline__ float lut_gather(uint32_t w, int k, int m) {
    uint32_t a;
    switch (k) {   // constant after unrolling
        case 0: asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(w)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(w)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(w)); break;
        default: asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(w)); break;
    }
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(a + 1024u * (uint32_t)m));
}

// byte table (row m at LDS byte 256 * m): the address is the code byte itself, the row rides in the immediate
__device__ __forceinline__ uint32_t lut_gather_u8(uint32_t w, int k, int m) {
    uint32_t a;
    switch (k) {   // constant after unrolling
        case 0: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(a) : "v"(w)); break;
        case 1: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(a) : "v"(w)); break;
        case 2: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(a) : "v"(w)); break;
        default: asm("v_lshrrev_b32 %0, 24, %1" : "=v"(a) : "v"(w)); break;
    }
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t*>((uintptr_t)(a + 256u * (uint32_t)m));
}

// a second byte table at LDS byte BASE
template <int BASE>
__device__ __forceinline__ uint32_t lut_gather_u8_at(uint32_t w, int k, int m) {
    uint32_t a;
    switch (k) {   // constant after unrolling
        case 0: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(a) : "v"(w)); break;
        case 1: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(a) : "v"(w)); break;
        case 2: asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(a) : "v"(w)); break;
        default: asm("v_lshrrev_b32 %0, 24, %1" : "=v"(a) : "v"(w)); break;
    }
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t*>((uintptr_t)(a + (uint32_t)BASE + 256u * (uint32_t)m));
}

// LUT entry e = tid + 256 * i goes to LDS with ds_write_addtid_b32: address = M0 + offset + 4 * lane, no address
// VGPR, 2 LDS cycles per wave instruction instead of the 4 of ds_write_b32 (MI355X_MICROARCH.md, LDS table).
// M0 = LDS address of the wave's 256-byte segment of table row 0 (lut_store_begin, once per LUT: an SALU write
// of M0 needs a wait state before an LDS add-TID instruction and the hazard recogniser does not look inside
// asm statements, hence the s_nop); row i rides in the 16-bit offset field, 1024 * i, i <= 63.  The stores are
// invisible to the compiler's wait counters, hence the explicit wait before the barrier (lut_store_done).
// Nothing else in these kernels touches M0 (no LDS-DMA, no movrel): check `grep m0` on the disassembly when
// the toolchain changes.
__device__ __forceinline__ void lut_store_begin(uint32_t m0_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(m0_base) : "memory");
}
template <int MT, int I>
__device__ __forceinline__ void lut_store_one(float v) {
    if constexpr (MT >= 64) {   // a 64 KB LUT reaches past the 16 bits of M0 / the offset field
        extern __shared__ float s_lut_plain[];
        s_lut_plain[threadIdx.x + 256 * I] = v;
    } else {
        asm volatile("ds_write_addtid_b32 %0 offset:%1" : : "v"(v), "n"(1024 * I) : "memory");
    }
}
// rows 0 .. MT-1 of one LUT: f(i) is the entry of row i for this thread
template <int MT, typename F, int... I>
__device__ __forceinline__ void lut_store_rows(F&& f, std::integer_sequence<int, I...>) {
    (lut_store_one<MT, I>(f(I)), ...);
}
// one row at a compile-time byte offset (the dual-query LUT of scan_lm.hip has 2 * MT rows)
template <int OFF>
__device__ __forceinline__ void lut_store_imm(float v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    asm volatile("ds_write_addtid_b32 %0 offset:%1" : : "v"(v), "n"(OFF) : "memory");
}
template <int BASE, int ROW_BYTES, typename F, int... I>
__device__ __forceinline__ void lut_store_rows_at(F&& f, std::integer_sequence<int, I...>) {
    (lut_store_imm<ROW_BYTES * (BASE + I)>(f(I)), ...);
}
__device__ __forceinline__ void lut_store_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace gh
