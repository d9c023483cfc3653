// What the instruction kinds the decode kernel is made of cost on gfx950: dependent chains against independent streams,
// one wavefront alone on its SIMD against 2-4 wavefronts per SIMD (started together behind a barrier).  Cycles from
// s_memtime around 256 instructions; per wavefront, and per SIMD (span of its wavefronts / instructions they issued).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_latency.hip -o tools/build/valu_latency
//   tools/build/valu_latency <blocks> <wavefronts per block: 1, 4 = one per SIMD, 8, 16>      (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define CASE(name, init, body)                                                                         \
    __global__ void k_##name(uint64_t *out, uint32_t seed)                                             \
    {                                                                                                  \
        uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = seed | 1, f = seed + 0x8000, w = 0xFFFF, g = a + 9, h = a + 11; \
        init;                                                                                          \
        uint64_t t0, t1;                                                                               \
        __syncthreads();                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
        asm volatile(REP64(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w), "+v"(g), "+v"(h) : "v"(e), "v"(f) : "s20", "s21", "s22", "s23"); \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                    \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; } \
        if (a + b + c + d + w + g + h == 0x12345) out[0] = 0;                                          \
    }

// each body = 4 instructions
CASE(add_dep,    , "v_add_u32 %0, %0, %7\n v_add_u32 %0, %0, %7\n v_add_u32 %0, %0, %7\n v_add_u32 %0, %0, %7\n")
CASE(add_ind2,   , "v_add_u32 %0, %0, %7\n v_add_u32 %1, %1, %7\n v_add_u32 %0, %0, %7\n v_add_u32 %1, %1, %7\n")
CASE(add_ind4,   , "v_add_u32 %0, %0, %7\n v_add_u32 %1, %1, %7\n v_add_u32 %2, %2, %7\n v_add_u32 %3, %3, %7\n")
CASE(lshladd_dep,, "v_lshl_add_u32 %0, %0, 1, %7\n v_lshl_add_u32 %0, %0, 1, %7\n v_lshl_add_u32 %0, %0, 1, %7\n v_lshl_add_u32 %0, %0, 1, %7\n")
CASE(lshladd_ind4,, "v_lshl_add_u32 %0, %0, 1, %7\n v_lshl_add_u32 %1, %1, 1, %7\n v_lshl_add_u32 %2, %2, 1, %7\n v_lshl_add_u32 %3, %3, 1, %7\n")
CASE(sdwa_dep,   , "v_mul_i32_i24_sdwa %0, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                   "v_mul_i32_i24_sdwa %0, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
                   "v_mul_i32_i24_sdwa %0, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                   "v_mul_i32_i24_sdwa %0, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n")
CASE(sdwa_ind4,  , "v_mul_i32_i24_sdwa %0, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                   "v_mul_i32_i24_sdwa %1, sext(%1), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
                   "v_mul_i32_i24_sdwa %2, sext(%2), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                   "v_mul_i32_i24_sdwa %3, sext(%3), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n")
CASE(pk_dep,     , "v_pk_add_u16 %0, %0, %7\n v_pk_sub_i16 %0, %0, %8 clamp\n v_pk_add_u16 %0, %0, %7\n v_pk_sub_i16 %0, %0, %8 clamp\n")
CASE(pk_ind4,    , "v_pk_add_u16 %0, %0, %7\n v_pk_sub_i16 %1, %1, %8 clamp\n v_pk_add_u16 %2, %2, %7\n v_pk_sub_i16 %3, %3, %8 clamp\n")
CASE(perm_dep,   , "v_perm_b32 %0, %0, %7, %8\n v_perm_b32 %0, %0, %7, %8\n v_perm_b32 %0, %0, %7, %8\n v_perm_b32 %0, %0, %7, %8\n")
CASE(perm_ind4,  , "v_perm_b32 %0, %0, %7, %8\n v_perm_b32 %1, %1, %7, %8\n v_perm_b32 %2, %2, %7, %8\n v_perm_b32 %3, %3, %7, %8\n")
CASE(min3_dep,   , "v_min3_u16 %4, %4, %0, %1\n v_min3_u16 %4, %4, %2, %3\n v_min3_u16 %4, %4, %0, %1\n v_min3_u16 %4, %4, %2, %3\n")
CASE(min3_ind2,  , "v_min3_u16 %4, %4, %0, %1\n v_min3_u16 %5, %5, %2, %3\n v_min3_u16 %4, %4, %0, %1\n v_min3_u16 %5, %5, %2, %3\n")
CASE(mul24_dep,  , "v_mul_i32_i24 %0, %0, %7\n v_mul_i32_i24 %0, %0, %7\n v_mul_i32_i24 %0, %0, %7\n v_mul_i32_i24 %0, %0, %7\n")
CASE(mad24_dep,  , "v_mad_i32_i24 %0, %0, %7, %8\n v_mad_i32_i24 %0, %0, %7, %8\n v_mad_i32_i24 %0, %0, %7, %8\n v_mad_i32_i24 %0, %0, %7, %8\n")
CASE(mad24_ind4, , "v_mad_i32_i24 %0, %0, %7, %8\n v_mad_i32_i24 %1, %1, %7, %8\n v_mad_i32_i24 %2, %2, %7, %8\n v_mad_i32_i24 %3, %3, %7, %8\n")
CASE(dot2_dep,   , "v_dot2_i32_i16 %0, %0, %7, %8\n v_dot2_i32_i16 %0, %0, %7, %8\n v_dot2_i32_i16 %0, %0, %7, %8\n v_dot2_i32_i16 %0, %0, %7, %8\n")
CASE(dot2_ind4,  , "v_dot2_i32_i16 %0, %0, %7, %8\n v_dot2_i32_i16 %1, %1, %7, %8\n v_dot2_i32_i16 %2, %2, %7, %8\n v_dot2_i32_i16 %3, %3, %7, %8\n")
CASE(cndmask_dep,, "v_cndmask_b32 %0, %0, %7, vcc\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %0, %0, %7, vcc\n v_cndmask_b32 %0, %0, %8, vcc\n")
CASE(alignbit_dep,, "v_alignbit_b32 %0, %0, %7, %8\n v_alignbit_b32 %0, %0, %7, %8\n v_alignbit_b32 %0, %0, %7, %8\n v_alignbit_b32 %0, %0, %7, %8\n")
// the rotate's chain as the compiler emits it: 2 products -> sub -> lshl_add -> perm -> pk_add
CASE(chain_mix,  , "v_mul_i32_i24_sdwa %1, sext(%0), sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n"
                   "v_lshl_add_u32 %1, %1, 1, %8\n v_perm_b32 %1, %1, %1, %8\n v_pk_add_u16 %0, %1, %0\n")
// scalar instructions between vector ones
CASE(salu_mix,   , "v_add_u32 %0, %0, %7\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %7\n s_add_u32 s21, s21, 1\n")
CASE(salu_only,  , "s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n")
CASE(salu_ind,   , "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")


CASE(min3u32_dep,, "v_min3_u32 %4, %4, %0, %1\n v_min3_u32 %4, %4, %2, %3\n v_min3_u32 %4, %4, %0, %1\n v_min3_u32 %4, %4, %2, %3\n")
CASE(minu16_vop2,, "v_min_u16 %4, %4, %0\n v_min_u16 %4, %4, %1\n v_min_u16 %4, %4, %2\n v_min_u16 %4, %4, %3\n")
CASE(pkmin_dep,  , "v_pk_min_u16 %4, %4, %0\n v_pk_min_u16 %4, %4, %1\n v_pk_min_u16 %4, %4, %2\n v_pk_min_u16 %4, %4, %3\n")
CASE(cnd_e64,    , "v_cndmask_b32_e64 %0, %0, %7, s[22:23]\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %7, s[22:23]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n")
CASE(cnd_vcc_ind,, "v_cndmask_b32 %0, %0, %7, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %7, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n")
CASE(add_sgpr,   , "v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3\n")
CASE(cmp_vcc,    , "v_cmp_gt_u32 vcc, %0, %7\n v_cmp_gt_u32 vcc, %1, %7\n v_cmp_gt_u32 vcc, %2, %7\n v_cmp_gt_u32 vcc, %3, %7\n")
CASE(cmp_e64,    , "v_cmp_gt_u32_e64 s[22:23], %0, %7\n v_cmp_gt_u32_e64 s[22:23], %1, %7\n v_cmp_gt_u32_e64 s[22:23], %2, %7\n v_cmp_gt_u32_e64 s[22:23], %3, %7\n")
CASE(cmp_cnd,    , "v_cmp_gt_u32 vcc, %0, %7\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_u32 vcc, %2, %7\n v_cndmask_b32 %3, %3, %8, vcc\n")
CASE(and_or,     , "v_and_or_b32 %0, %0, %7, %8\n v_and_or_b32 %1, %1, %7, %8\n v_and_or_b32 %2, %2, %7, %8\n v_and_or_b32 %3, %3, %7, %8\n")
CASE(bfe,        , "v_bfe_i32 %0, %0, 3, 16\n v_bfe_i32 %1, %1, 3, 16\n v_bfe_i32 %2, %2, 3, 16\n v_bfe_i32 %3, %3, 3, 16\n")
CASE(ashr_vop2,  , "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n")
CASE(add3,       , "v_add3_u32 %0, %0, %7, %8\n v_add3_u32 %1, %1, %7, %8\n v_add3_u32 %2, %2, %7, %8\n v_add3_u32 %3, %3, %7, %8\n")
CASE(mov,        , "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
CASE(nop,        , "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
CASE(add_sdwa,   , "v_add_u32_sdwa %0, %0, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %1, %1, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                   "v_add_u32_sdwa %2, %2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n")
CASE(mul_lo_u32, , "v_mul_lo_u32 %0, %0, %7\n v_mul_lo_u32 %1, %1, %7\n v_mul_lo_u32 %2, %2, %7\n v_mul_lo_u32 %3, %3, %7\n")
CASE(readlane,   , "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s20, %2, 3\n v_readlane_b32 s21, %3, 5\n")
CASE(dpp_mov,    , "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
// a compare and the select that uses it, the mask in VCC (what the compiler emits) against in another SGPR pair
CASE(cmpcnd_e64, , "v_cmp_gt_u32_e64 s[22:23], %0, %7\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cmp_gt_u32_e64 s[22:23], %2, %7\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n")
CASE(vcc_salu,   , "v_cmp_gt_u32 vcc, %0, %7\n s_and_b64 s[20:21], vcc, exec\n v_cmp_gt_u32 vcc, %2, %7\n s_and_b64 s[22:23], vcc, exec\n")
// dependent LDS reads (the address of each is the value the one before returned): the LDS round trip of a lone wavefront
CASE(lds_chain,  __shared__ uint32_t tab[256]; tab[threadIdx.x & 255] = ((threadIdx.x + 1) & 255) * 4; __syncthreads(); a = (threadIdx.x & 255) * 4 + static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab)) * 0,
                 "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n")
CASE(lds_write,  __shared__ uint32_t tab[256]; tab[threadIdx.x & 255] = 0; __syncthreads(); a = (threadIdx.x & 255) * 4,
                 "ds_write_b16 %0, %1\n ds_write_b16 %0, %2 offset:2\n ds_write_b16 %0, %1\n ds_write_b16 %0, %2 offset:2\n")

// the first memory round trip of a kernel: every wavefront of the chip requests NLOADS x 16 bytes per lane of its own
// region at once (the decode kernel's prologue asks for about 3.5 KB per wavefront), cycles until all of it is there
template <int NLOADS>
__global__ void k_gload(uint64_t *out, const uint4 *src, uint32_t seed)
{
    const size_t base = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) + seed * 0;
    uint64_t t0, t1;
    uint4 v[NLOADS];
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int k = 0 ; k < NLOADS ; ++k)
        v[k] = src[base + static_cast<size_t>(k) * gridDim.x * blockDim.x];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0 ; k < NLOADS ; ++k)
        acc += v[k].x + v[k].w;
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(acc) :: "memory");
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (acc == 0x12345) out[0] = 0;
}

// straight-line code that is long for the instruction cache's prefetch: 2 048 eight-byte (resp. four-byte) instructions,
// executed once per wavefront -- do they still issue at the short sequences' rate?
#define REP512(x) REP64(x) REP64(x) REP64(x) REP64(x) REP64(x) REP64(x) REP64(x) REP64(x)
#define LONGCASE(name, body)                                                                           \
    __global__ void k_##name(uint64_t *out, uint32_t seed)                                             \
    {                                                                                                  \
        uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = seed | 1;         \
        uint64_t t0, t1;                                                                               \
        __syncthreads();                                                                               \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                    \
        asm volatile(REP512(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));                      \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                    \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; } \
        if (a + b + c + d == 0x12345) out[0] = 0;                                                      \
    }
LONGCASE(long8, "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4\n")
LONGCASE(long4, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")

typedef void (*Kern)(uint64_t *, uint32_t);
struct Entry { const char *name; Kern k; };

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;         // one wavefront per block
    const int waves = argc > 2 ? atoi(argv[2]) : 1;            // wavefronts per block (=> per CU; 4 = one per SIMD)
    uint64_t *d;
    hipMalloc(&d, sizeof(uint64_t) * blocks * 32);
    std::vector<uint64_t> h(blocks * 32);
#define E(n) { #n, k_##n }
    Entry es[] = { E(add_dep), E(add_ind2), E(add_ind4), E(lshladd_dep), E(lshladd_ind4), E(sdwa_dep), E(sdwa_ind4), E(pk_dep), E(pk_ind4),
                   E(perm_dep), E(perm_ind4), E(min3_dep), E(min3_ind2), E(mul24_dep), E(mad24_dep), E(mad24_ind4), E(dot2_dep), E(dot2_ind4),
                   E(cndmask_dep), E(alignbit_dep), E(chain_mix), E(salu_mix), E(salu_only), E(salu_ind),
                   E(min3u32_dep), E(minu16_vop2), E(pkmin_dep), E(cnd_e64), E(cnd_vcc_ind), E(add_sgpr), E(cmp_vcc), E(cmp_e64), E(cmp_cnd), E(and_or), E(bfe), E(ashr_vop2), E(add3), E(mov), E(nop),
                   E(cmpcnd_e64), E(vcc_salu), E(add_sdwa), E(mul_lo_u32), E(readlane), E(dpp_mov), E(lds_chain), E(lds_write) };
    printf("blocks %d, wavefronts per block %d; cycles per instruction (256 instructions timed), median over blocks\n", blocks, waves);
    for (const Entry &e : es)
    {
        for (int rep = 0 ; rep < 3 ; ++rep)
        {
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64 * waves), 0, 0, d, 12345u + rep);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, sizeof(uint64_t) * blocks * 32, hipMemcpyDeviceToHost);
        // per wavefront: its own cycles per instruction; per block: cycles the SIMD spent per instruction it issued
        std::vector<double> perWave, perSimd;
        for (int b = 0 ; b < blocks ; ++b)
        {
            uint64_t lo = ~0ull, hi = 0;
            for (int w = 0 ; w < waves ; ++w)
            {
                const uint64_t t0 = h[(b * 16 + w) * 2], t1 = h[(b * 16 + w) * 2 + 1];
                perWave.push_back((t1 - t0) / 256.0);
                lo = std::min(lo, t0); hi = std::max(hi, t1);
            }
            perSimd.push_back((hi - lo) / (256.0 * std::max(1, waves / 4)));
        }
        std::sort(perWave.begin(), perWave.end());
        std::sort(perSimd.begin(), perSimd.end());
        printf("%-14s per wavefront: median %6.2f  min %6.2f  max %6.2f | per SIMD (span / instructions of its wavefronts): median %6.2f\n", e.name,
               perWave[perWave.size() / 2], perWave.front(), perWave.back(), perSimd[perSimd.size() / 2]);
    }
    for (int which = 0 ; which < 2 ; ++which)
    {
        for (int rep = 0 ; rep < 3 ; ++rep)
        {
            if (which == 0) hipLaunchKernelGGL(k_long8, dim3(blocks), dim3(64 * waves), 0, 0, d, 777u + rep);
            else hipLaunchKernelGGL(k_long4, dim3(blocks), dim3(64 * waves), 0, 0, d, 777u + rep);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, sizeof(uint64_t) * blocks * 32, hipMemcpyDeviceToHost);
            std::vector<double> c;
            for (int b = 0 ; b < blocks ; ++b)
                for (int w = 0 ; w < waves ; ++w)
                    c.push_back(static_cast<double>(h[(b * 16 + w) * 2 + 1] - h[(b * 16 + w) * 2]) / 2048.0);
            std::sort(c.begin(), c.end());
            printf("%-14s launch %d: cycles per instruction over 2 048 straight-line instructions: median %5.2f  min %5.2f  max %5.2f\n",
                   which == 0 ? "long 8-byte" : "long 4-byte", rep, c[c.size() / 2], c.front(), c.back());
        }
    }
    {
        uint4 *src;
        const size_t n = static_cast<size_t>(blocks) * 64 * waves * 8;
        hipMalloc(&src, n * sizeof(uint4));
        hipMemset(src, 1, n * sizeof(uint4));
        auto report = [&](const char *name)
        {
            hipMemcpy(h.data(), d, sizeof(uint64_t) * blocks * 32, hipMemcpyDeviceToHost);
            std::vector<double> c;
            for (int b = 0 ; b < blocks ; ++b)
                for (int w = 0 ; w < waves ; ++w)
                    c.push_back(static_cast<double>(h[(b * 16 + w) * 2 + 1] - h[(b * 16 + w) * 2]));
            std::sort(c.begin(), c.end());
            printf("%-14s cycles until the data is there: median %6.0f  min %6.0f  max %6.0f\n", name, c[c.size() / 2], c.front(), c.back());
        };
        for (int rep = 0 ; rep < 4 ; ++rep) { hipLaunchKernelGGL(k_gload<1>, dim3(blocks), dim3(64 * waves), 0, 0, d, src, 0u); hipDeviceSynchronize(); }
        report("gload 1x16B");
        for (int rep = 0 ; rep < 4 ; ++rep) { hipLaunchKernelGGL(k_gload<4>, dim3(blocks), dim3(64 * waves), 0, 0, d, src, 0u); hipDeviceSynchronize(); }
        report("gload 4x16B");
        for (int rep = 0 ; rep < 4 ; ++rep) { hipLaunchKernelGGL(k_gload<8>, dim3(blocks), dim3(64 * waves), 0, 0, d, src, 0u); hipDeviceSynchronize(); }
        report("gload 8x16B");
        hipFree(src);
    }
    hipFree(d);
    return 0;
}
