// Dev-only: issue cost of the integer instructions the walk kernels are made of, relative to v_add_u32 (gfx950).  The fused walk
// kernel of the 2-hop configurations is bound by VALU issue (84 % busy); Philox2x32-10 is 20 multiplies per call.  Is a 32x32
// multiply full rate or quarter rate here, and does v_mad_u64_u32 (hi AND lo in one instruction) cost one multiply or two?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/build/valu_probe && tools/build/valu_probe
// 2,048 blocks x 256 lanes (8 waves per SIMD), every wave runs `iters` x 32 copies of the instruction on 4 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))
template <int OP>
__global__ __launch_bounds__(256) void probe(int iters, uint32_t *out) {
    uint32_t a = threadIdx.x * 2654435761u + 1u, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b + 999u;
    unsigned long long q0 = a, q1 = b;
    const uint32_t m = 0xD256D193u;
    for (int k = 0; k < iters; ++k) {
        if (OP == 0) { REP32(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 1) { REP32(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 2) { REP32(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 3) { REP32(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 4) { REP32(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %2, %3, 0\n v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %2, %3, 0" : "+v"(q0), "+v"(q1) : "v"(a), "v"(m) : "vcc");) }
        if (OP == 5) { REP32(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 6) { REP32(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 7) { REP32(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
        if (OP == 8) { REP32(asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0" : "+v"(q0), "+v"(q1));) }
        if (OP == 9) { REP32(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 10) { REP32(asm volatile("v_bfe_u32 %0, %0, 3, 20\n v_bfe_u32 %1, %1, 3, 20\n v_bfe_u32 %2, %2, 3, 20\n v_bfe_u32 %3, %3, 3, 20" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 11) { REP32(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 12) { REP32(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 14) { REP32(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s20", "s21");) }
        if (OP == 15) { REP32(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
        if (OP == 16) { REP32(asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %4\n v_cmp_lt_u32_e64 s[22:23], %1, %4\n v_cmp_lt_u32_e64 s[24:25], %2, %4\n v_cmp_lt_u32_e64 s[26:27], %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 17) { REP32(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");) }
        if (OP == 18) { REP32(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 19) { REP32(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 20) { REP32(asm volatile("v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_xad_u32 %2, %2, %4, %3\n v_xad_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 21) { REP32(asm volatile("v_add_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 22) { REP32(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 23) { REP32(asm volatile("v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "s20");) }
        if (OP == 24) { REP32(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 25) { REP32(asm volatile("v_and_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_lshrrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (OP == 13) { REP32(asm volatile("s_add_u32 s20, s20, 1\n s_xor_b32 s21, s21, s20\n s_add_u32 s22, s22, 1\n s_xor_b32 s23, s23, s22" ::: "s20", "s21", "s22", "s23", "scc");) }
    }
    if ((a ^ b ^ c ^ d ^ (uint32_t)q0 ^ (uint32_t)(q1 >> 32)) == 0x12345u) out[0] = a;
}
template <int OP>
static double run(int iters, uint32_t *d_out) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    probe<OP><<<2048, 256>>>(iters, d_out);
    hipEventRecord(a);
    probe<OP><<<2048, 256>>>(iters, d_out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    uint32_t *d_out;
    hipMalloc(&d_out, 4);
    const int iters = 256;
    const char *names[] = {"v_add_u32", "v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_lshl_add_u32", "v_cndmask_b32",
                           "v_lshl_add_u64", "v_mad_u32_u24", "v_bfe_u32", "v_add3_u32", "v_add_u32_dpp row_shr", "s_add/s_xor (SALU)", "v_cndmask_b32_e64 sgpr", "v_cmp_lt_u32 vcc", "v_cmp_lt_u32_e64 sgpr",
                           "v_cmp vcc + v_cndmask", "v_mov_b32", "v_min_u32", "v_xad_u32", "add/mul_lo alternating", "v_add_u32 literal", "v_add_u32 sgpr",
                           "v_lshlrev_b32", "and/or/sub/lshr mix"};
    double t[26];
    t[0] = run<0>(iters, d_out), t[1] = run<1>(iters, d_out), t[2] = run<2>(iters, d_out), t[3] = run<3>(iters, d_out), t[4] = run<4>(iters, d_out);
    t[5] = run<5>(iters, d_out), t[6] = run<6>(iters, d_out), t[7] = run<7>(iters, d_out), t[8] = run<8>(iters, d_out), t[9] = run<9>(iters, d_out);
    t[10] = run<10>(iters, d_out), t[11] = run<11>(iters, d_out), t[12] = run<12>(iters, d_out), t[13] = run<13>(iters, d_out);
    t[14] = run<14>(iters, d_out), t[15] = run<15>(iters, d_out), t[16] = run<16>(iters, d_out), t[17] = run<17>(iters, d_out), t[18] = run<18>(iters, d_out);
    t[19] = run<19>(iters, d_out), t[20] = run<20>(iters, d_out), t[21] = run<21>(iters, d_out), t[22] = run<22>(iters, d_out), t[23] = run<23>(iters, d_out);
    t[24] = run<24>(iters, d_out), t[25] = run<25>(iters, d_out);
    // instructions per wave = iters * 128; waves per SIMD = 2048 * 4 / (256 * 4) = 8
    for (int i = 0; i < 26; ++i) {
        const double inst_per_simd = (double)iters * 128.0 * 8.0;
        printf("%-24s %8.3f ms   %6.2f x v_add_u32   (%.2f ns per wave-instruction per SIMD)\n", names[i], t[i], t[i] / t[0], t[i] * 1e6 / inst_per_simd);
    }
    return 0;
}
