// K2x4: the GRU recurrence (H = 256) with a row group's gate columns SPLIT OVER FOUR CUs.
//
// gru_seq16_kernel (gru16.hip) keeps a 16-row group on ONE CU: per step 144 MFMAs per wave on that CU's matrix pipes
// (2.2 us) and 60 of a wave's 96 W_hh fragments re-streamed from L2 (480 KB per step and CU) -- 4 us per step with few CUs
// busy, 5.8 us with all of them, and a 1024-row document batch occupies 64 of the chip's 256 CUs.  Here a row group is a
// TEAM of four workgroups (members), one per CU; member m owns hidden units [64 m, 64 m + 64) of all three gates:
//   * its quarter of W_hh -- 192 gate columns x 256 k, fp16 hi + lo = 196 KB -- lives in VGPRs for the whole sequence
//     (4 waves x 48 fragments x 4 registers; one wave per SIMD with the whole register file): NOTHING is streamed per step;
//   * per step a member runs 72 MFMAs per wave (0.55 us), the gate math of its 64 units x 16 rows, and then the members
//     exchange their new hidden states -- as the fp16 hi | lo pairs the next step's A operand is made of -- through
//     device memory: 8-byte {hi|lo, tag = step + 1} GRANULES written with sc1 (write-through) stores and swept with sc1
//     loads until every tag matches (MI355X_MICROARCH.md "visibility", cdna_hip_programming.md Guideline 16, form R2: the
//     data is the flag; no fence, no separate flag round trip; placement-independent -- members on one XCD are faster,
//     members on different XCDs are still correct).  Two parities of granule slots: a member can only be ONE step ahead
//     of the slowest (it needs everybody's step-s state to produce step s + 1), so the slot it overwrites has been read.
//   * arithmetic, operand order and rounding are gru_seq16_kernel's: a column's accumulator sees the same products in the
//     same order, so outputs, stash and final states are BIT-IDENTICAL to that kernel (tests/test_encoder_gpu.py).
// Co-residency: the grid is at most one workgroup per CU (the host only takes this path when 4 x row groups x directions
// <= CUs); members of a team that is not resident yet are waited for with a BOUNDED sweep: a wave that exhausts its budget
// raises bit 2 (value 4) of the call's status word and the team leaves the step loop -- it never spins forever.
#include "encoder.h"
#include "sgemm.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int X4_H = 256;
constexpr int X4_NK = X4_H / 32;          // k-steps of 32
constexpr int X4_LDH = X4_H + 8;          // fp16 elements per row of an h image (as gru_seq16_kernel)
constexpr int X4_IMG = 16 * X4_LDH * 2;   // bytes of one (hi or lo) image
constexpr int X4_LDS = 4 * X4_IMG;        // [buffer 2][hi, lo]
constexpr int X4_H_SHIFT = 10;            // h is scaled by 2^10 before the split (gru16.hip: H_SHIFT)
constexpr int X4_REGION = 16 * 64 * 8;    // one member's granules of one parity: [row 16][unit 64] x 8 B
constexpr size_t X4_TEAM_BYTES = 2 * 4 * (size_t)X4_REGION; // [parity][member]

__device__ __forceinline__ h8 frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

struct GruSplitParams {
    GruParams g;
    char *xch;         // [dir][team] x X4_TEAM_BYTES, zeroed before the launch (tag 0 = nothing published)
    int32_t *status;   // nullable: bit 2 (value 4) = an exchange sweep timed out
    int nteams;
    unsigned spin_max; // sweeps a wave makes for one step's granules before it gives up
};

__global__ __launch_bounds__(256, 1) void gru_seq16x4_kernel(GruSplitParams sp)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int &abort_flag = *(int *)(lds + X4_LDS); // (no static __shared__: it would shift the dynamic region off 16-byte alignment)
    const GruParams &p = sp.g;
    // blocks b and b + 8 share an XCD under round-robin dispatch (a speed bonus, never relied on): the four members of a
    // team are 8 apart inside a 32-block chunk
    const int chunk = blockIdx.x >> 5, r32 = blockIdx.x & 31;
    const int m = r32 >> 3, team = chunk * 8 + (r32 & 7);
    if (team >= sp.nteams)
        return;
    const GruDir d = p.dir[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = team * ENC_RB;
    constexpr int H = X4_H, H3 = 3 * X4_H;

    int len_e[4], off_e[4], rid_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int br = row0 + kq * 4 + e;
        rid_e[e] = br < p.B ? p.perm[br] : -1;
        len_e[e] = rid_e[e] >= 0 ? p.len[rid_e[e]] : 0;
        off_e[e] = rid_e[e] >= 0 ? p.tok_off[rid_e[e]] : 0;
    }
    int steps = max(max(len_e[0], len_e[1]), max(len_e[2], len_e[3])); // the row group's longest row
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));

    const int u16 = 4 * m + w;      // this wave's 16 hidden units: [16 u16, 16 u16 + 16) of every gate
    const int unit = 16 * u16 + j;  // this lane's unit
    const int ex = tt_pow2_exponent(*d.wmax);
    const float up = ldexpf(1.0f, X4_H_SHIFT + ex), down = ldexpf(1.0f, -(X4_H_SHIFT + ex));
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        bias[g] = d.b_hh[g * H + unit] * up;
    for (int i = tid; i < X4_LDS / 4; i += 256)
        ((int *)lds)[i] = 0; // h_0 = 0 in both buffers, both parts
    if (tid == 0)
        abort_flag = 0;
    float hreg[4] = {0, 0, 0, 0};

    // ---- this wave's 48 fragments of W_hh, resident for the whole sequence.  The packed order is gru16_pack's (wave pw of
    // 32 units, fragment f = 12 s + 4 g + 2 part + ct): this wave's units are (pw, ct) = (u16 >> 1, u16 & 1) ----
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)d.wp + (size_t)(u16 >> 1) * 96 * 1024), 0, 96 * 1024, 0x00020000);
    const int loff = lane * 16 + (u16 & 1) * 1024;
    h8 wreg[X4_NK][3][2]; // [k-step][gate][hi, lo]
#pragma unroll
    for (int s2 = 0; s2 < X4_NK; ++s2)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                wreg[s2][g][part] = frag_load(wsrc, loff, (12 * s2 + 4 * g + 2 * part) * 1024);

    // ---- the team's granule slots ----
    char *const xteam = sp.xch + ((size_t)blockIdx.y * sp.nteams + team) * X4_TEAM_BYTES;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xteam, 0, (int)X4_TEAM_BYTES, 0x00020000);
    // send: lanes j and j ^ 1 hold units (2 q, 2 q + 1) of rows 4 kq .. 4 kq + 3; the even lane publishes rows e = 0, 1, the
    // odd lane rows e = 2, 3, each as ONE 16-byte store {unit 2q | tag | unit 2q+1 | tag}
    const bool odd = j & 1;
    const int send_row = kq * 4 + (odd ? 2 : 0);
    const int send_off = (send_row * 64 + 16 * w + (j & ~1)) * 8; // + 512 for the second row
    // (the resident fragments have landed: without this the compiler keeps vmcnt waits for them INSIDE the step loop, where
    //  they would also wait for the next step's prefetched projections and this step's stores)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();

    auto gi_load = [&](int s, float (&gv)[3][4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const size_t tk = (size_t)(off_e[e] + (a ? t : 0)); // (a valid token even when the row is done)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gv[g][e] = d.gi[tk * H3 + g * H + unit];
        }
    };
    float giv[3][4], gnx[3][4];
    if (steps > 0)
        gi_load(0, giv);
    __builtin_amdgcn_s_waitcnt(0x0F70); // (once: otherwise every step's gate math waits for the NEXT step's prefetch)

    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        bool act[4];
        size_t tok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tok[e] = (size_t)(off_e[e] + (act[e] ? t : 0));
        }
        if (s + 1 < steps)
            gi_load(s + 1, gnx); // in flight under this step's MFMAs and hand-off
        f32x4v acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
            acc[g] = (f32x4v){bias[g], bias[g], bias[g], bias[g]};

        const char *img = lds + cur * 2 * X4_IMG + j * (X4_LDH * 2) + kq * 16;
        h8 a_hi[2], a_lo[2]; // by k-step parity; the next k-step's A fragments are read under this one's nine MFMAs
        a_hi[0] = *(const h8 *)(img);
        a_lo[0] = *(const h8 *)(img + X4_IMG);
#pragma unroll
        for (int s2 = 0; s2 < X4_NK; ++s2) {
            if (s2 + 1 < X4_NK) {
                a_hi[(s2 + 1) & 1] = *(const h8 *)(img + (s2 + 1) * 64);
                a_lo[(s2 + 1) & 1] = *(const h8 *)(img + X4_IMG + (s2 + 1) * 64);
            }
            // per column tile and k-step: hi*hi, lo*hi, hi*lo -- gru_seq16_kernel's order
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[s2][g][0], acc[g], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 1)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], wreg[s2][g][0], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[s2][g][1], acc[g], 0, 0, 0);
#endif
        }

        char *nimg = lds + (cur ^ 1) * 2 * X4_IMG;
        unsigned pk[4]; // fp16 hi | lo << 16 of this lane's four new states
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = tt_fast_sigmoid(giv[0][e] + acc[0][e] * down);
            const float z = tt_fast_sigmoid(giv[1][e] + acc[1][e] * down);
            const float ghn = acc[2][e] * down;
            const float n = tt_fast_tanh(giv[2][e] + r * ghn);
            const float hn = (hreg[e] - n) * z + n;
            if (act[e]) {
                hreg[e] = hn;
                if (d.out_seq)
                    d.out_seq[tok[e] * p.out_ld + d.out_col0 + unit] = hn;
                if (d.gates) {
                    float *gs = d.gates + tok[e] * 4 * H + unit;
                    gs[0] = r;
                    gs[H] = z;
                    gs[2 * H] = n;
                    gs[3 * H] = ghn;
                }
            }
            const float hs = hreg[e] * (float)(1 << X4_H_SHIFT);
            const _Float16 hi = (_Float16)hs;
            const _Float16 lo = (_Float16)(hs - (float)hi);
            _Float16 *dst = (_Float16 *)nimg + (kq * 4 + e) * X4_LDH + unit;
            dst[0] = hi;
            dst[X4_IMG / 2] = lo;
            pk[e] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
        if (s + 1 < steps) {
            const unsigned tag = (unsigned)s + 1u;
            const int par = s & 1;
            // ---- publish: swap halves with the neighbour lane, then two 16-byte write-through stores ----
            const unsigned g0 = __shfl_xor(odd ? pk[0] : pk[2], 1), g1 = __shfl_xor(odd ? pk[1] : pk[3], 1);
            const u32x4 v0 = odd ? (u32x4){g0, tag, pk[2], tag} : (u32x4){pk[0], tag, g0, tag};
            const u32x4 v1 = odd ? (u32x4){g1, tag, pk[3], tag} : (u32x4){pk[1], tag, g1, tag};
            const int sbase = (par * 4 + m) * X4_REGION + send_off;
            __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, sbase, 0, 16);       // aux 16 = sc1
            __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, sbase + 512, 0, 16); // the next row
            // ---- sweep the other three members' granules until every tag is this step's ----
            u32x4 got[3][2];
            bool ok = false;
            unsigned spins = 0;
            while (true) {
                ok = true;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int om = (m + 1 + o) & 3;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        got[o][c] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, (par * 4 + om) * X4_REGION + (tid + 256 * c) * 16, 0, 16);
                        ok = ok && got[o][c].y == tag && got[o][c].w == tag;
                    }
                }
                if (__all(ok))
                    break;
                if (++spins > sp.spin_max) { // (wave-uniform: spins is)
                    if (lane == 0) {
                        abort_flag = 1;
                        if (sp.status)
                            atomicOr(sp.status, 4);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // chunk (tid + 256 c) of a member's region: row = chunk >> 5, units 2 (chunk & 31), + 1
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int om = (m + 1 + o) & 3;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ch = tid + 256 * c;
                    unsigned *dst = (unsigned *)(nimg + ((ch >> 5) * X4_LDH + 64 * om + 2 * (ch & 31)) * 2);
                    const unsigned a = got[o][c].x, b = got[o][c].z;
                    dst[0] = (a & 0xffffu) | (b << 16);
                    dst[X4_IMG / 4] = (a >> 16) | (b & 0xffff0000u);
                }
            }
        }
        __syncthreads();
        if (abort_flag)
            break;
        cur ^= 1;
        if (s + 1 < steps) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    giv[g][e] = gnx[g][e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (rid_e[e] >= 0)
            d.h_final[(size_t)rid_e[e] * H + unit] = hreg[e];
}

int device_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    return cus;
}

} // namespace

// TT_GRU_SPLIT=0 keeps every recurrence on gru_seq16_kernel (A/B, and the reference point of the bit-identity tests)
// (read at every call, so that one process can compare the two kernels)
static bool split_enabled()
{
    const char *e = getenv("TT_GRU_SPLIT");
    return !(e && e[0] == '0');
}

size_t gru16x4_xch_bytes(int B, int H, int ndir)
{
    if (H != X4_H || B <= 0 || B > 1024) // (more than 64 row groups never fit one workgroup per CU four times over)
        return 0;
    return (size_t)ndir * ((B + ENC_RB - 1) / ENC_RB) * X4_TEAM_BYTES;
}

// one workgroup per CU at most, so that every member of every team is resident (one wave per SIMD with the whole
// register file: nothing else fits on a CU beside one of these workgroups)
bool gru16x4_usable(int B, int H, int ndir)
{
    if (!split_enabled() || H != X4_H || B <= 0)
        return false;
    const int nteams = (B + ENC_RB - 1) / ENC_RB;
    return (nteams + 7) / 8 * 32 * ndir <= device_cus();
}

int gru16x4_launch(const GruParams &gp, int ndir, void *xch, int32_t *status, hipStream_t st)
{
    GruSplitParams sp;
    sp.g = gp;
    sp.xch = (char *)xch;
    sp.status = status;
    sp.nteams = (gp.B + ENC_RB - 1) / ENC_RB;
    sp.spin_max = 1u << 19; // ~0.5 s of sweeps: a partner that is merely waiting for a CU arrives long before that
    TT_RC_CHECK(tt_zero_async(xch, gru16x4_xch_bytes(gp.B, gp.H, ndir), st));
    static bool attr_done = false;
    if (!attr_done) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16x4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X4_LDS + 16));
        attr_done = true;
    }
    hipLaunchKernelGGL(gru_seq16x4_kernel, dim3((sp.nteams + 7) / 8 * 32, ndir), dim3(256), X4_LDS + 16, st, sp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
