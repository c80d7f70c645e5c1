// Prototype + microbenchmark of the "register tile" predict walk (round 4): lane = row, the row's F features live in a FIXED
// bank of VGPRs that the compiler never touches (amdgpu_num_vgpr caps its own allocation; the bank is named in clobber lists so
// the kernel descriptor covers it), every level of an oblivious tree is `s_set_gpr_idx_idx f ; v_cmp_gt_f32 mask, v[bank + M0], t`
// (the feature index is wave-uniform: VGPR-relative addressing through M0 instead of an LDS read per level), leaf values are
// gathered from LDS with four ds_read_b64 per tree and applied with v_pk_fma_f32 in tree order.
//   mode 0: row loads only (each lane loads its own 512-byte row with 32 global_load_dwordx4)
//   mode 1: resident ensemble (T <= 72 trees in LDS), persistent blocks
//   mode 2: grouped ensemble (TT trees per group, double-buffered), one tile per wave
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/rowreg_bench.hip -o /tmp/rowreg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- register map -------------------------------------------------------------------------------------------------------
// compiler: v0..v95 (amdgpu_num_vgpr(48): the cap counts VGPR + AGPR halves, so 48 -> 96 architectural VGPRs)
// tile:     v96..v223 (128 features)
// temps:    v224 leaf a, v225 leaf b, v[226:233] values a, v[234:241] values b
#define CLOB_TILE \
    "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
    "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
    "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143", \
    "v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
    "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175", \
    "v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191", \
    "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207", \
    "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223"
#define CLOB_TEMPS "v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241"
#define CLOB_SGPR \
    "s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59", \
    "s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79", \
    "s80","s81","s82","s83","s84","s85","s86","s87"

// 32 x 16 bytes of this lane's row -> v96..v223
#define LD4(r, o) "global_load_dwordx4 v[" #r ":" #r "+3], %0, off offset:" #o "\n\t"
__device__ __forceinline__ void load_row_tile(const float *row) {
    asm volatile(
        LD4(96, 0) LD4(100, 16) LD4(104, 32) LD4(108, 48) LD4(112, 64) LD4(116, 80) LD4(120, 96) LD4(124, 112)
        LD4(128, 128) LD4(132, 144) LD4(136, 160) LD4(140, 176) LD4(144, 192) LD4(148, 208) LD4(152, 224) LD4(156, 240)
        LD4(160, 256) LD4(164, 272) LD4(168, 288) LD4(172, 304) LD4(176, 320) LD4(180, 336) LD4(184, 352) LD4(188, 368)
        LD4(192, 384) LD4(196, 400) LD4(200, 416) LD4(204, 432) LD4(208, 448) LD4(212, 464) LD4(216, 480) LD4(220, 496)
        :: "v"(row) : "memory", CLOB_TILE);
}

// One level: M0[7:0] <- feature index, mask <- (x[f] > t).  s_set_gpr_idx_on does the same as _idx and switches the mode on.
#define LVL_ON(fs, ts, m)  "s_set_gpr_idx_on s" #fs ", gpr_idx(SRC0)\n\tv_cmp_gt_f32_e64 s[" #m ":" #m "+1], v96, s" #ts "\n\t"
#define LVL(fs, ts, m)     "s_set_gpr_idx_idx s" #fs "\n\tv_cmp_gt_f32_e64 s[" #m ":" #m "+1], v96, s" #ts "\n\t"
#define ADDC0(l, m)        "v_addc_co_u32_e64 v" #l ", vcc, 0, 0, s[" #m ":" #m "+1]\n\t"
#define ADDC(l, m)         "v_addc_co_u32_e64 v" #l ", vcc, v" #l ", v" #l ", s[" #m ":" #m "+1]\n\t"

// Walk `pairs` pairs of depth-6 trees: records at cp (12 dwords per tree: (feature, threshold bits) x 6), values in LDS at byte
// offset vb (2048 bytes per tree: [slice 4][leaf 64][2 floats]); p += -lr * v in tree order.
__device__ __forceinline__ void walk_pairs(const int32_t *cp, uint32_t vb, int pairs, f32x2 &p0, f32x2 &p1, f32x2 &p2, f32x2 &p3,
                                           f32x2 n0, f32x2 n1, f32x2 n2, f32x2 n3) {
    const uint32_t cpl = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(cp)), cph = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(cp) >> 32);
    asm volatile(
        "s_mov_b32 s38, %[cpl]\n\t"
        "s_mov_b32 s39, %[cph]\n\t"
        "s_load_dwordx16 s[40:55], s[38:39], 0x0\n\t"
        "s_load_dwordx8 s[56:63], s[38:39], 0x40\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        LVL_ON(40, 41, 64) LVL(42, 43, 66) LVL(44, 45, 68) LVL(46, 47, 70) LVL(48, 49, 72) LVL(50, 51, 74)
        LVL(52, 53, 76) LVL(54, 55, 78) LVL(56, 57, 80) LVL(58, 59, 82) LVL(60, 61, 84) LVL(62, 63, 86)
        "s_set_gpr_idx_off\n\t"
        "s_load_dwordx16 s[40:55], s[38:39], 0x60\n\t"
        "s_load_dwordx8 s[56:63], s[38:39], 0xa0\n\t"
        ADDC0(224, 64) ADDC0(225, 76) ADDC(224, 66) ADDC(225, 78) ADDC(224, 68) ADDC(225, 80)
        ADDC(224, 70) ADDC(225, 82) ADDC(224, 72) ADDC(225, 84) ADDC(224, 74) ADDC(225, 86)
        "v_lshl_add_u32 v224, v224, 3, %[vb]\n\t"
        "v_lshl_add_u32 v225, v225, 3, %[vb]\n\t"
        "ds_read_b64 v[226:227], v224\n\t"
        "ds_read_b64 v[228:229], v224 offset:512\n\t"
        "ds_read_b64 v[230:231], v224 offset:1024\n\t"
        "ds_read_b64 v[232:233], v224 offset:1536\n\t"
        "ds_read_b64 v[234:235], v225 offset:2048\n\t"
        "ds_read_b64 v[236:237], v225 offset:2560\n\t"
        "ds_read_b64 v[238:239], v225 offset:3072\n\t"
        "ds_read_b64 v[240:241], v225 offset:3584\n\t"
        "s_add_u32 s38, s38, 96\n\t"
        "s_addc_u32 s39, s39, 0\n\t"
        "s_add_u32 %[vb], %[vb], 4096\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[p0], %[n0], v[226:227], %[p0]\n\t"
        "v_pk_fma_f32 %[p1], %[n1], v[228:229], %[p1]\n\t"
        "v_pk_fma_f32 %[p2], %[n2], v[230:231], %[p2]\n\t"
        "v_pk_fma_f32 %[p3], %[n3], v[232:233], %[p3]\n\t"
        "v_pk_fma_f32 %[p0], %[n0], v[234:235], %[p0]\n\t"
        "v_pk_fma_f32 %[p1], %[n1], v[236:237], %[p1]\n\t"
        "v_pk_fma_f32 %[p2], %[n2], v[238:239], %[p2]\n\t"
        "v_pk_fma_f32 %[p3], %[n3], v[240:241], %[p3]\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [vb] "+s"(vb), [n] "+s"(pairs)
        : [cpl] "s"(cpl), [cph] "s"(cph), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [n3] "v"(n3)
        : "memory", "vcc", "scc", CLOB_TILE, CLOB_TEMPS, CLOB_SGPR);
}


// ---- software-pipelined walk (v2): the leaf values of step s are in flight while the leaves of step s + 1 are searched; record
// buffers R0 = s[28:51], R1 = s[52:75] (one step = two trees = 24 dwords), masks s[76:99], cond pointer s[26:27].  Every
// s_waitcnt lgkmcnt(0) only meets requests that are at least one step old.
#define CLOB_SGPR2 \
    "s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49", \
    "s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73", \
    "s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99"
#if defined(EXP_NOCMP)
#define XLVL(fs, ts, m)     ""
#elif defined(EXP_NOIDX)
#define XLVL(fs, ts, m)     "s_nop 0\n\tv_cmp_gt_f32_e64 s[" #m ":" #m "+1], v96, s" #ts "\n\t"
#elif defined(EXP_CMPONLY)
#define XLVL(fs, ts, m)     "v_cmp_gt_f32_e64 s[" #m ":" #m "+1], v96, s" #ts "\n\t"
#elif defined(EXP_IDXONLY)
#define XLVL(fs, ts, m)     "s_set_gpr_idx_idx s" #fs "\n\t"
#else
#define XLVL(fs, ts, m)     LVL(fs, ts, m)
#endif
#define CMP_R0 LVL_ON(28, 29, 76) XLVL(30, 31, 78) XLVL(32, 33, 80) XLVL(34, 35, 82) XLVL(36, 37, 84) XLVL(38, 39, 86) \
               XLVL(40, 41, 88) XLVL(42, 43, 90) XLVL(44, 45, 92) XLVL(46, 47, 94) XLVL(48, 49, 96) XLVL(50, 51, 98) "s_set_gpr_idx_off\n\t"
#define CMP_R1 LVL_ON(52, 53, 76) XLVL(54, 55, 78) XLVL(56, 57, 80) XLVL(58, 59, 82) XLVL(60, 61, 84) XLVL(62, 63, 86) \
               XLVL(64, 65, 88) XLVL(66, 67, 90) XLVL(68, 69, 92) XLVL(70, 71, 94) XLVL(72, 73, 96) XLVL(74, 75, 98) "s_set_gpr_idx_off\n\t"
#ifdef EXP_NOADDC
#define ADDC_AB "v_mov_b32 v224, 0\n\tv_mov_b32 v225, 0\n\t"
#else
#define ADDC_AB ADDC0(224, 76) ADDC0(225, 88) ADDC(224, 78) ADDC(225, 90) ADDC(224, 80) ADDC(225, 92) \
                ADDC(224, 82) ADDC(225, 94) ADDC(224, 84) ADDC(225, 96) ADDC(224, 86) ADDC(225, 98)
#endif
#define LOAD_R0P(off) "s_load_dwordx16 s[28:43], s[26:27], " #off "\n\ts_load_dwordx8 s[44:51], s[26:27], " #off "+0x40\n\t"
#define LOAD_R1P(off) "s_load_dwordx16 s[52:67], s[26:27], " #off "\n\ts_load_dwordx8 s[68:75], s[26:27], " #off "+0x40\n\t"
#ifdef EXP_NOSMEM
#define LOAD_R0(off) ""
#define LOAD_R1(off) ""
#else
#define LOAD_R0(off) "s_load_dwordx16 s[28:43], s[26:27], " #off "\n\ts_load_dwordx8 s[44:51], s[26:27], " #off "+0x40\n\t"
#define LOAD_R1(off) "s_load_dwordx16 s[52:67], s[26:27], " #off "\n\ts_load_dwordx8 s[68:75], s[26:27], " #off "+0x40\n\t"
#endif
#ifdef EXP_NOFMA
#define FMA_AB ""
#else
#define FMA_AB \
        "v_pk_fma_f32 %[p0], %[n0], v[226:227], %[p0]\n\tv_pk_fma_f32 %[p1], %[n1], v[228:229], %[p1]\n\t" \
        "v_pk_fma_f32 %[p2], %[n2], v[230:231], %[p2]\n\tv_pk_fma_f32 %[p3], %[n3], v[232:233], %[p3]\n\t" \
        "v_pk_fma_f32 %[p0], %[n0], v[234:235], %[p0]\n\tv_pk_fma_f32 %[p1], %[n1], v[236:237], %[p1]\n\t" \
        "v_pk_fma_f32 %[p2], %[n2], v[238:239], %[p2]\n\tv_pk_fma_f32 %[p3], %[n3], v[240:241], %[p3]\n\t"
#endif
#ifdef EXP_NODS
#define DSREAD_AB "s_add_u32 %[vb], %[vb], 4096\n\t"
#else
#define DSREAD_AB \
        "v_lshl_add_u32 v224, v224, 3, %[vb]\n\tv_lshl_add_u32 v225, v225, 3, %[vb]\n\t" \
        "ds_read_b64 v[226:227], v224\n\tds_read_b64 v[228:229], v224 offset:512\n\t" \
        "ds_read_b64 v[230:231], v224 offset:1024\n\tds_read_b64 v[232:233], v224 offset:1536\n\t" \
        "ds_read_b64 v[234:235], v225 offset:2048\n\tds_read_b64 v[236:237], v225 offset:2560\n\t" \
        "ds_read_b64 v[238:239], v225 offset:3072\n\tds_read_b64 v[240:241], v225 offset:3584\n\t" \
        "s_add_u32 %[vb], %[vb], 4096\n\t"
#endif
__device__ __forceinline__ void walk_pairs2(const int32_t *cp, uint32_t vb, int steps, f32x2 &p0, f32x2 &p1, f32x2 &p2, f32x2 &p3,
                                            f32x2 n0, f32x2 n1, f32x2 n2, f32x2 n3) {
    const uint32_t cpl = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(cp)), cph = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(cp) >> 32);
    asm volatile(
        "s_mov_b32 s26, %[cpl]\n\t"
        "s_mov_b32 s27, %[cph]\n\t"
        LOAD_R0P(0x0) LOAD_R1P(0x60)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        CMP_R0 LOAD_R0(0xc0) ADDC_AB DSREAD_AB
        "s_cmp_eq_u32 %[n], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "1:\n\t"
        CMP_R1 ADDC_AB
        "s_waitcnt lgkmcnt(0)\n\t"
        LOAD_R1(0x120)
        FMA_AB DSREAD_AB
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_eq_u32 %[n], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        CMP_R0 ADDC_AB
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_add_u32 s26, s26, 0xc0\n\t"
        "s_addc_u32 s27, s27, 0\n\t"
        LOAD_R0(0xc0)
        FMA_AB DSREAD_AB
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        FMA_AB
        : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [vb] "+s"(vb), [n] "+s"(steps)
        : [cpl] "s"(cpl), [cph] "s"(cph), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [n3] "v"(n3)
        : "memory", "vcc", "scc", CLOB_TILE, CLOB_TEMPS, CLOB_SGPR2);
}
#ifdef WALK2
#define walk_pairs walk_pairs2
#endif

__device__ __forceinline__ void wait_tile() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory", CLOB_TILE); }
struct Coef { float lr[8]; float bias[8]; };

// mode 0/1: persistent blocks, the whole ensemble's leaf values resident in LDS
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void k_resident(const float *__restrict__ obs, int n, const int32_t *__restrict__ cond,
                                                                                       const float *__restrict__ vals, int T, Coef cf,
                                                                                       float *__restrict__ out, int reps) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (MODE == 1) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(vals);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid; i < T * 128; i += 256) dst[i] = src[i];
        __syncthreads();
    }
    const int n_tiles = (n + 63) >> 6;
    const int wstride = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += wstride) {
        const int row = min(tile * 64 + lane, n - 1);
        load_row_tile(obs + static_cast<size_t>(row) * 128);
        f32x2 p0 = {cf.bias[0], cf.bias[1]}, p1 = {cf.bias[2], cf.bias[3]}, p2 = {cf.bias[4], cf.bias[5]}, p3 = {cf.bias[6], cf.bias[7]};
        if (MODE == 1) {
            const f32x2 n0 = {-cf.lr[0], -cf.lr[1]}, n1 = {-cf.lr[2], -cf.lr[3]}, n2 = {-cf.lr[4], -cf.lr[5]}, n3 = {-cf.lr[6], -cf.lr[7]};
            wait_tile();
            for (int r = 0; r < reps; ++r) walk_pairs(cond, 0u, T >> 1, p0, p1, p2, p3, n0, n1, n2, n3);
        } else {
            float s;
            asm volatile("s_waitcnt vmcnt(0)\n\tv_add_f32 %0, v96, v223\n\tv_add_f32 %0, %0, v160" : "=v"(s) :: CLOB_TILE);
            p0.x += s;
        }
        if (tile * 64 + lane < n) {
            f32x4 *o = reinterpret_cast<f32x4 *>(out + static_cast<size_t>(tile * 64 + lane) * 8);
            o[0] = f32x4{p0.x, p0.y, p1.x, p1.y};
            o[1] = f32x4{p2.x, p2.y, p3.x, p3.y};
        }
    }
}

// mode 2: one tile per wave, groups of TT trees double-buffered in LDS
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void k_grouped(const float *__restrict__ obs, int n, const int32_t *__restrict__ cond,
                                                                                      const float *__restrict__ vals, int T, int TT, Coef cf,
                                                                                      float *__restrict__ out) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x * 4 + wave;
    const int row = min(tile * 64 + lane, n - 1);
    load_row_tile(obs + static_cast<size_t>(row) * 128);
    wait_tile();
    const int n_groups = (T + TT - 1) / TT;
    const int g4 = TT * 128;                 // float4 per group
    const f32x4 *src = reinterpret_cast<const f32x4 *>(vals);
    f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
    // groups 0 and 1 (the mirror is padded: whole groups can always be copied)
    for (int i = tid; i < g4; i += 256) dst[i] = src[i];
    if (n_groups > 1) for (int i = tid; i < g4; i += 256) dst[g4 + i] = src[g4 + i];
    f32x2 p0 = {cf.bias[0], cf.bias[1]}, p1 = {cf.bias[2], cf.bias[3]}, p2 = {cf.bias[4], cf.bias[5]}, p3 = {cf.bias[6], cf.bias[7]};
    const f32x2 n0 = {-cf.lr[0], -cf.lr[1]}, n1 = {-cf.lr[2], -cf.lr[3]}, n2 = {-cf.lr[4], -cf.lr[5]}, n3 = {-cf.lr[6], -cf.lr[7]};
    __syncthreads();
    for (int g = 0; g < n_groups; ++g) {
        const int tn = min(TT, T - g * TT);
        // the values of group g + 2 travel to registers while group g is walked (8 float4 per thread cover TT <= 16)
        f32x4 r[8];
        const bool more = g + 2 < n_groups;
        if (more) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = tid + u * 256; r[u] = src[static_cast<size_t>(g + 2) * g4 + min(i, g4 - 1)]; }
        }
        walk_pairs(cond + static_cast<size_t>(g) * TT * 12, static_cast<uint32_t>((g & 1) * g4 * 16), tn >> 1, p0, p1, p2, p3, n0, n1, n2, n3);
        __syncthreads();   // everybody is done with buffer g & 1
        if (more) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = tid + u * 256; if (i < g4) dst[(g & 1) * g4 + i] = r[u]; }
        }
        // (the stores become visible to the other waves at the next iteration's barrier, before group g + 2 is walked)
    }
    if (tile * 64 + lane < n) {
        f32x4 *o = reinterpret_cast<f32x4 *>(out + static_cast<size_t>(tile * 64 + lane) * 8);
        o[0] = f32x4{p0.x, p0.y, p1.x, p1.y};
        o[1] = f32x4{p2.x, p2.y, p3.x, p3.y};
    }
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const int T = argc > 2 ? atoi(argv[2]) : 28;          // even
    const int n = argc > 3 ? atoi(argv[3]) : (1 << 20);
    const int bpc = argc > 4 ? atoi(argv[4]) : 2;         // blocks per CU (mode 0/1)
    const int TT = argc > 5 ? atoi(argv[5]) : 16;
    const int reps_walk = argc > 6 ? atoi(argv[6]) : 1;
    const int F = 128, D = 8, MD = 6, LS = 64;
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> obs(static_cast<size_t>(n) * F);
    for (auto &v : obs) v = nd(rng);
    const int Tpad = T + 32;
    std::vector<int32_t> cond(static_cast<size_t>(Tpad) * 2 * MD, 0);
    std::vector<float> vals(static_cast<size_t>(Tpad) * LS * D + 65536, 0.f);
    for (int t = 0; t < Tpad; ++t)
        for (int d = 0; d < MD; ++d) {
            float th = t < T ? nd(rng) * 0.7f : INFINITY;
            int32_t tb; memcpy(&tb, &th, 4);
            cond[(t * MD + d) * 2] = t < T ? static_cast<int>(rng() % F) : 0;
            cond[(t * MD + d) * 2 + 1] = tb;
        }
    for (int t = 0; t < T; ++t)
        for (int leaf = 0; leaf < LS; ++leaf)
            for (int j = 0; j < D; ++j) vals[static_cast<size_t>(t) * LS * D + ((j / 2) * LS + leaf) * 2 + (j % 2)] = nd(rng);
    Coef cf;
    for (int j = 0; j < D; ++j) { cf.lr[j] = j < 7 ? 0.1f : 0.01f; cf.bias[j] = 0.25f * j; }
    float *d_obs, *d_vals, *d_out; int32_t *d_cond;
    CK(hipMalloc(&d_obs, obs.size() * 4)); CK(hipMalloc(&d_vals, vals.size() * 4)); CK(hipMalloc(&d_cond, cond.size() * 4));
    CK(hipMalloc(&d_out, static_cast<size_t>(n) * D * 4));
    CK(hipMemcpy(d_obs, obs.data(), obs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cond, cond.data(), cond.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, static_cast<size_t>(n) * D * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds_res = static_cast<size_t>(T) * 2048, lds_grp = static_cast<size_t>(2) * TT * 2048;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resident<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resident<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_grouped), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k_resident<0>, dim3(256 * bpc), dim3(256), 0, 0, d_obs, n, d_cond, d_vals, T, cf, d_out, 1);
        else if (mode == 1) hipLaunchKernelGGL(k_resident<1>, dim3(256 * bpc), dim3(256), lds_res, 0, d_obs, n, d_cond, d_vals, T, cf, d_out, reps_walk);
        else hipLaunchKernelGGL(k_grouped, dim3((n + 255) / 256), dim3(256), lds_grp, 0, d_obs, n, d_cond, d_vals, T, TT, cf, d_out);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = static_cast<double>(n) * (F + D) * 4;
    const int T_eff = T * reps_walk;
    printf("mode %d T %d n %d bpc %d TT %d: %.4f ms  %.2f TB/s  %.3e row-trees/s  clk/(64 rows,tree,CU) %.1f\n", mode, T, n, bpc, TT, ms, bytes / ms * 1e-9,
           static_cast<double>(n) * T_eff / (ms * 1e-3), ms * 1e-3 * 2.4e9 / (static_cast<double>(n) / 64 / 256 * (T_eff > 0 ? T_eff : 1)));
    if (mode >= 1 && reps_walk == 1) {
        std::vector<float> out(static_cast<size_t>(n) * D);
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, checked = 0;
        const int stepr = T > 100 ? 61 : 7;
        for (int r = 0; r < n; r += stepr) {
            float p[8];
            for (int j = 0; j < D; ++j) p[j] = cf.bias[j];
            for (int t = 0; t < (T & ~1); ++t) {
                int leaf = 0;
                for (int d = 0; d < MD; ++d) {
                    float th; memcpy(&th, &cond[(t * MD + d) * 2 + 1], 4);
                    leaf = 2 * leaf + (obs[static_cast<size_t>(r) * F + cond[(t * MD + d) * 2]] > th ? 1 : 0);
                }
                for (int j = 0; j < D; ++j) p[j] = fmaf(-cf.lr[j], vals[static_cast<size_t>(t) * LS * D + ((j / 2) * LS + leaf) * 2 + (j % 2)], p[j]);
            }
            for (int j = 0; j < D; ++j) { ++checked; if (memcmp(&p[j], &out[static_cast<size_t>(r) * D + j], 4) != 0) { if (bad < 5) printf("row %d out %d: %g vs %g\n", r, j, p[j], out[static_cast<size_t>(r) * D + j]); ++bad; } }
        }
        printf("check: %zu of %zu values differ (bitwise)\n", bad, checked);
    }
    return 0;
}
