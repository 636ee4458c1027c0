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
// Since round 3 a member is EIGHT waves, two per SIMD (gru_seq16x4p_kernel / gru_bwd16x4p_kernel below): a lone wave issues one
// instruction per ~8.7 clocks whatever its kind, two waves on a SIMD each do (tools/experiments/valu_rate.hip), and a step of
// these kernels is little more than its instruction count.  The four-wave members are compiled into the comparison build only
// (-DTT_AB; TT_GRU_SPLIT=4 / TT_GRU_SPLIT_BWD=4 there) as the reference the tests compare the eight-wave ones with; everything
// said here about the protocol holds for both.
// Co-residency: the grid is at most one workgroup per CU (ONE launch is only taken when 4 x row groups x directions <= CUs; a host
// that keeps two calls in flight orders their recurrences or gives the smaller one TT_ENC_ONE_WORKGROUP: include/tt.h,
// trainer._towers_in_flight); members of a team that is not resident yet are waited for with a BOUNDED sweep: a wave that
// exhausts its budget raises bit 2 (value 4) of the call's status word and the team leaves the step loop -- it never spins forever.
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
[[maybe_unused]] constexpr int X4_LDS = 4 * X4_IMG; // [buffer 2][hi, lo] (these three: the four-wave member of the comparison build)
constexpr int X4_H_SHIFT = 10;            // h is scaled by 2^10 before the split (gru16.hip: H_SHIFT)
constexpr int X4_REGION = 16 * 64 * 8;    // one member's granules of one parity: [row 16][unit 64] x 8 B
constexpr size_t X4_TEAM_BYTES = 2 * 4 * (size_t)X4_REGION; // [parity][member]

__device__ __forceinline__ h8 frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

// Which XCD this workgroup runs on (hardware register, 0-7).
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}

// Progressive back-off of a sweep that did not find its tags: the first retries come quickly (partners in step arrive
// within a microsecond), later ones seldom -- a member whose partner has no CU yet would otherwise pull 24 KB through
// the fabric every microsecond for as long as it waits.
__device__ __forceinline__ void sweep_backoff(unsigned spins)
{
    if (spins < 8)
        __builtin_amdgcn_s_sleep(1);
    else if (spins < 32)
        __builtin_amdgcn_s_sleep(8);
    else
        __builtin_amdgcn_s_sleep(64);
}

constexpr int X4_HEADER = 256; // per team, behind its granule slots: one {XCC id, tag} granule per member

// Do the four members of this team share an XCD (= one L2)?  Every member publishes its XCC id (sc1: placement-independent)
// and reads all four; all members compute the same answer from the same four words.  If they do, the step granules are
// written with PLAIN stores: the line stays in the XCD's L2, where the partners' sc1 loads (which bypass only their own L1)
// find it -- no trip through the fabric.  If they do not, the stores are sc1 (write-through) and every read comes from
// memory: slower, equally correct.  Returns 1 / 0, or -1 when a partner did not show up within the budget.
__device__ __forceinline__ int team_shares_xcd(__amdgpu_buffer_rsrc_t xsrc, int header_off, int m, int tid, unsigned spin_max, int *lds_word)
{
    if (tid == 0) {
        const unsigned long long g = ((unsigned long long)0x5843u << 32) | (xcc_id() + 1u); // tag 'XC', value id + 1 (never 0)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, g), xsrc,
                                              header_off + 8 * m, 0, 16);
        int res = -1;
        for (unsigned spins = 0; spins <= spin_max; ++spins) {
            unsigned id[4];
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const __attribute__((ext_vector_type(2))) unsigned v = __builtin_amdgcn_raw_buffer_load_b64(xsrc, header_off + 8 * q, 0, 16);
                id[q] = v.x;
                ok = ok && v.y == 0x5843u;
            }
            if (ok) {
                res = (id[0] == id[1] && id[1] == id[2] && id[2] == id[3]) ? 1 : 0;
                break;
            }
            sweep_backoff(spins);
        }
        *lds_word = res;
    }
    __syncthreads();
    return *lds_word;
}

// lane ^ 1 and a 16-lane rotate as DPP moves (VALU, no LDS crossbar round trip as __shfl_xor's ds_bpermute)
__device__ __forceinline__ unsigned swap1(unsigned x)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ float row_ror(float x) // lane i of a 16-lane row takes lane (i + N) % 16's value
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x120 + N, 0xF, 0xF, true));
}

struct GruSplitParams {
    GruParams g;
    char *xch;         // [dir][team] x X4_TEAM_BYTES, zeroed before the launch (tag 0 = nothing published)
    int32_t *status;   // nullable: bit 2 (value 4) = an exchange sweep timed out
    int nteams;
    unsigned spin_max; // sweeps a wave makes for one step's granules before it gives up
};

#ifdef TT_AB // the four-wave member: superseded by gru_seq16x4p_kernel, kept in the comparison build as its bit-identity reference
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
    char *const xteam = sp.xch + ((size_t)blockIdx.y * sp.nteams + team) * (X4_TEAM_BYTES + X4_HEADER);
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xteam, 0, (int)X4_TEAM_BYTES + X4_HEADER, 0x00020000);
    // send: lanes j and j ^ 1 hold units (2 q, 2 q + 1) of rows 4 kq .. 4 kq + 3; the even lane publishes rows e = 0, 1, the
    // odd lane rows e = 2, 3, each as ONE 16-byte store {unit 2q | tag | unit 2q+1 | tag}
    const bool odd = j & 1;
    const int send_row = kq * 4 + (odd ? 2 : 0);
    const int send_off = (send_row * 64 + 16 * w + (j & ~1)) * 8; // + 512 for the second row
    const int same_xcd = team_shares_xcd(xsrc, (int)X4_TEAM_BYTES, m, tid, sp.spin_max, &abort_flag + 1);
    if (same_xcd < 0) {
        if (tid == 0 && sp.status)
            atomicOr(sp.status, 4);
        steps = 0; // (h_final = 0 is written below; the status bit tells the caller the outputs are invalid)
    }
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
        float sv_r[4], sv_z[4], sv_n[4], sv_g[4]; // what the training stash keeps of this step (stored BEHIND the hand-off)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = tt_fast_sigmoid(giv[0][e] + acc[0][e] * down);
            const float z = tt_fast_sigmoid(giv[1][e] + acc[1][e] * down);
            const float ghn = acc[2][e] * down;
            const float n = tt_fast_tanh(giv[2][e] + r * ghn);
            const float hn = (hreg[e] - n) * z + n;
            if (act[e])
                hreg[e] = hn;
            sv_r[e] = r;
            sv_z[e] = z;
            sv_n[e] = n;
            sv_g[e] = ghn;
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
            const unsigned g0 = swap1(odd ? pk[0] : pk[2]), g1 = swap1(odd ? pk[1] : pk[3]);
            const u32x4 v0 = odd ? (u32x4){g0, tag, pk[2], tag} : (u32x4){pk[0], tag, g0, tag};
            const u32x4 v1 = odd ? (u32x4){g1, tag, pk[3], tag} : (u32x4){pk[1], tag, g1, tag};
            const int sbase = (par * 4 + m) * X4_REGION + send_off;
            if (same_xcd) { // the partners read this XCD's L2
                __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, sbase, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, sbase + 512, 0, 0); // the next row
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, sbase, 0, 16);      // aux 16 = sc1 (write-through)
                __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, sbase + 512, 0, 16);
            }
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
                sweep_backoff(spins);
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
        // The step's bulk stores go out only now: vector memory operations complete in issue order, so in front of the
        // hand-off they would stand between the sweep's loads and the registers those loads fill (20 stores per lane).
        // From here they drain under the next step's MFMAs.
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (act[e]) {
                if (d.out_seq)
                    d.out_seq[tok[e] * p.out_ld + d.out_col0 + unit] = hreg[e];
                if (d.gates) {
                    float *gs = d.gates + tok[e] * 4 * H + unit;
                    gs[0] = sv_r[e];
                    gs[H] = sv_z[e];
                    gs[2 * H] = sv_n[e];
                    gs[3 * H] = sv_g[e];
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
#endif // TT_AB

// ------------------------------------------------------------------ the forward recurrence with TWO waves per SIMD
// (see gru_bwd16x4p_kernel below for why: a lone wave issues one instruction per ~8.7 clocks, and the step is its instruction
// count.)  A member is eight waves; the pair (w, w + 4) shares unit slice u16 = 4 m + w:
//   * the pair splits the GATES: half 0 multiplies for r and z (48 MFMAs, 32 resident W fragments), half 1 for n (24 MFMAs, 16
//     fragments).  A column's accumulator chain -- bias, then k-steps 0 .. 7, three products each -- stays inside one wave, so
//     outputs, stash and final states are BIT-IDENTICAL to gru_seq16_kernel and to the four-wave gru_seq16x4_kernel, and the
//     matrix pipe is busy for 72 MFMAs per SIMD and step as before.  (A first build handed the chain itself from half 0 (k-steps
//     0 .. 3) to half 1 (4 .. 7) through LDS: two hops and two barriers per step around the MFMAs, 2 360 clocks for 1 152 of
//     matrix work.)
//   * then ONE exchange through LDS (barrier X2): half 0 gives the r and z sums of rows 2, 3, half 1 the n sums of rows 0, 1, and
//     each half does gate math, fp16 split, image write, publish (ONE 16-byte store per lane: the even lane of a pair row 2 kh,
//     the odd lane row 2 kh + 1), stash / output stores and the next step's projections for ITS two rows of the lane's four;
//   * the sweep of the other members' granules is one 16-byte load per member and thread (barrier X3 closes the step).
//
// LDS image of h (fp16 hi / lo, two buffers), laid out for the MFMA A-fragment reads.  Lane (j = row, kq) of a 16x16x32 fragment
// reads 16 bytes = units 32 s2 + 8 kq .. + 8 of row j with ds_read_b128, which the LDS serves in four groups of 16 lanes that
// each hold ALL 16 rows at two neighbouring kq ({0-3, 12-15} at one, {4-11} at the other: MI355X_MICROARCH.md, LDS).  A
// row-major image ([row][unit], 528-byte rows: round 3) puts kq 16 bytes = one bank slot further, so in every group one row of
// the second set lands on a slot of the first: 2-way, 8 LDS cycles per read instead of 4 -- 128 LDS cycles per k-step for the
// CU's eight waves against 72 of MFMA, SQ_LDS_BANK_CONFLICT 40 % of SQ_LDS_IDX_ACTIVE (profiles/r04_a_pmc_sq_train_kernels.txt).
// Here the image is [kq plane][row][k-step] x 16 bytes: the kq term is a multiple of the 256-byte bank row (no slot shift) and a
// row is 9 slots (odd: the 16 rows of a group fall on 16 different slots) -- conflict-free reads; the own-unit stores (2 bytes)
// are 2-way, which a store does not pay for, and the granule unpack stores are conflict-free with xf_chunk's chunk order.
constexpr int XF_ROWB = 8 * 16 + 16;      // bytes of one row in one plane: 8 k-steps x 16 B + one slot of padding
constexpr int XF_PLANE = 16 * XF_ROWB;    // 2304 = 9 x 256
constexpr int XF_IMG = 4 * XF_PLANE;      // bytes of one (hi or lo) image
constexpr int XF_IMGS = 4 * XF_IMG;       // [buffer 2][hi, lo]
constexpr int XF_RZ = 4 * 64 * 4 * 4;    // [w][lane][r, z sums of rows 2, 3] half 0 -> half 1
constexpr int XF_N = 4 * 64 * 2 * 4;     // [w][lane][n sums of rows 0, 1] half 1 -> half 0
constexpr int XF_LDS = XF_IMGS + XF_RZ + XF_N; // + 16 for the abort word

// Order of the 512 16-byte chunks (= two granules: units 2 p, 2 p + 1 of one row) in a member's exchange region.  The consumer's
// thread t loads chunk t (the sweep stays one contiguous 8 KB read per member) and stores it into the image: with this order
// the 64 stores of a wave are 16 rows x 4 dwords of ONE plane (32 banks per half-wave: conflict-free); a publishing wave still
// writes 128-byte runs (the four pairs of a fragment lane group, rows 2 kh and 2 kh + 1).
__device__ __forceinline__ int xf_chunk(int row, int p)
{
    return (p & 3) | ((row & 7) << 2) | ((row >> 3) << 5) | (((p >> 2) & 3) << 6) | ((p >> 4) << 8);
}

#ifdef TT_X4_DBG // a measuring build: s_memtime clocks per phase of the step, printed every 8th launch
__device__ unsigned long long x4_dbg[16];
#define X4_T(i) do { __builtin_amdgcn_sched_barrier(0); tm[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define X4_T(i) do { } while (0)
#endif
// GATHER: projections from the projected table (GruParams::gi_ids, gru16.hip): the id of a row's token at step s + 2 is asked
// for at the top of step s, right behind the loads of step s + 1's projections that use the id fetched one step earlier.
template <bool GATHER>
__global__ __launch_bounds__(512, 1) void gru_seq16x4p_kernel(GruSplitParams sp)
{
#ifdef TT_X4_DBG
    unsigned long long tm[7], tacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int &abort_flag = *(int *)(lds + XF_LDS);
    static_assert(XF_PLANE % 256 == 0 && (XF_ROWB / 16) % 2 == 1, "plane stride a multiple of the bank row, row stride an odd number of 16-B slots");
    const GruParams &p = sp.g;
    const int chunk = blockIdx.x >> 5, r32 = blockIdx.x & 31;
    const int m = r32 >> 3, team = chunk * 8 + (r32 & 7);
    if (team >= sp.nteams)
        return;
    const GruDir d = p.dir[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3, kh = wv >> 2;
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = team * ENC_RB;
    constexpr int H = X4_H, H3 = 3 * X4_H;

    int len_e[2], off_e[2], rid_e[2], steps = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { // (all four rows of the lane for the step count, two of them kept)
        const int br = row0 + kq * 4 + e;
        const int rid = br < p.B ? p.perm[br] : -1;
        const int len = rid >= 0 ? p.len[rid] : 0;
        steps = max(steps, len);
        if ((e >> 1) == kh) {
            rid_e[e & 1] = rid;
            len_e[e & 1] = len;
            off_e[e & 1] = rid >= 0 ? p.tok_off[rid] : 0;
        }
    }
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));

    const int u16 = 4 * m + w;      // the pair's 16 hidden units: [16 u16, 16 u16 + 16) of every gate
    const int unit = 16 * u16 + j;  // this lane's unit
    const int ex = tt_pow2_exponent(*d.wmax);
    const float up = ldexpf(1.0f, X4_H_SHIFT + ex), down = ldexpf(1.0f, -(X4_H_SHIFT + ex));
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        bias[g] = d.b_hh[g * H + unit] * up;
    for (int i = tid; i < XF_IMGS / 4; i += 512)
        ((int *)lds)[i] = 0; // h_0 = 0 in both buffers, both parts
    if (tid == 0)
        abort_flag = 0;
    float hreg[2] = {0, 0};
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    f32x4v *const xrz = (f32x4v *)(lds + XF_IMGS) + (w * 64 + lane);
    f32x2v *const xn = (f32x2v *)(lds + XF_IMGS + XF_RZ) + (w * 64 + lane);

    // ---- this wave's fragments of W_hh: gates r, z (half 0) or n (half 1), all eight k-steps; gru16_pack's order as in
    //      gru_seq16x4_kernel (fragment 12 s + 4 g + 2 part + ct of wave u16 >> 1) ----
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)d.wp + (size_t)(u16 >> 1) * 96 * 1024), 0, 96 * 1024, 0x00020000);
    const int loff = lane * 16 + (u16 & 1) * 1024;
    h8 wreg[X4_NK][2][2]; // [k-step][gate - 2 kh (half 1 uses index 0 only)][hi, lo]
#pragma unroll
    for (int s2 = 0; s2 < X4_NK; ++s2)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                if (kh == 0 || gg == 0)
                    wreg[s2][gg][part] = frag_load(wsrc, loff + (12 * s2 + 4 * (2 * kh + gg) + 2 * part) * 1024, 0);

    char *const xteam = sp.xch + ((size_t)blockIdx.y * sp.nteams + team) * (X4_TEAM_BYTES + X4_HEADER);
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xteam, 0, (int)X4_TEAM_BYTES + X4_HEADER, 0x00020000);
    const bool odd = j & 1;
    // the 16-byte chunk (two granules: units 2 p, 2 p + 1 of one row) this lane publishes, in xf_chunk's order
    const int send_off = xf_chunk(kq * 4 + 2 * kh + (odd ? 1 : 0), 8 * w + (j >> 1)) * 16;
    // where this lane's own unit sits in an image row (bytes): k-step u16 >> 1, fragment lane group 2 (u16 & 1) + (j >> 3)
    const int own_off = (2 * (u16 & 1) + (j >> 3)) * XF_PLANE + (u16 >> 1) * 16 + (j & 7) * 2;
    // the chunk this thread unpacks from every other member's region: chunk tid = (row urow, pair upr)
    const int urow = ((tid >> 2) & 7) | (((tid >> 5) & 1) << 3), upr = (tid & 3) | (((tid >> 6) & 3) << 2) | ((tid >> 8) << 4);
    const int unpack_off = ((upr >> 2) & 3) * XF_PLANE + urow * XF_ROWB + (upr >> 4) * 16 + 4 * (upr & 3); // + 32 om (two k-steps per member)
    const int same_xcd = team_shares_xcd(xsrc, (int)X4_TEAM_BYTES, m, tid, sp.spin_max, &abort_flag + 1);
    if (same_xcd < 0) {
        if (tid == 0 && sp.status)
            atomicOr(sp.status, 4);
        steps = 0; // (h_final = 0 is written below; the status bit tells the caller the outputs are invalid)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the resident fragments have landed (no such waits inside the loop)
    __syncthreads();

    int nrow[2] = {0, 0}; // GATHER: the projected-table rows of the step gi_load is asked for next
    auto tok_at = [&](int e, int s) {
        const bool a = s < len_e[e];
        const int t = d.reverse ? len_e[e] - 1 - s : s;
        return off_e[e] + (a ? t : 0); // (a valid token even when the row is done)
    };
    auto id_load = [&](int s) {
        if constexpr (GATHER) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                nrow[e] = p.gi_ids[tok_at(e, s)];
        }
    };
    auto gi_load = [&](int s, float (&gv)[3][2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const size_t tk = GATHER ? (size_t)min((unsigned)nrow[e], p.gi_rows - 1u) : (size_t)tok_at(e, s);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gv[g][e] = d.gi[tk * H3 + g * H + unit];
        }
    };
    float giv[3][2], gnx[3][2];
    id_load(0);
    if (steps > 0)
        gi_load(0, giv);
    id_load(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);

    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        bool act[2];
        size_t tok[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tok[e] = (size_t)(off_e[e] + (act[e] ? t : 0));
        }
        X4_T(0);
        if (s + 1 < steps) {
            gi_load(s + 1, gnx); // in flight under this step's MFMAs and hand-off
            id_load(s + 2);      // (GATHER) consumed at the top of the next step
        }
        const char *img = lds + cur * 2 * XF_IMG + kq * XF_PLANE + j * XF_ROWB; // (row j, fragment lane group kq: conflict-free b128 reads)
        h8 a_hi[2], a_lo[2]; // by k-step parity
        a_hi[0] = *(const h8 *)(img);
        a_lo[0] = *(const h8 *)(img + XF_IMG);
        f32x4v acc[2]; // half 0: r, z; half 1: n (index 0)
        acc[0] = (f32x4v){bias[2 * kh], bias[2 * kh], bias[2 * kh], bias[2 * kh]};
        acc[1] = (f32x4v){bias[1], bias[1], bias[1], bias[1]};
        auto run = [&](auto ngc) {
            constexpr int NG = decltype(ngc)::value;
#pragma unroll
            for (int s2 = 0; s2 < X4_NK; ++s2) {
                if (s2 + 1 < X4_NK) {
                    a_hi[(s2 + 1) & 1] = *(const h8 *)(img + (s2 + 1) * 16);
                    a_lo[(s2 + 1) & 1] = *(const h8 *)(img + XF_IMG + (s2 + 1) * 16);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[s2][g][0], acc[g], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 1)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], wreg[s2][g][0], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[s2][g][1], acc[g], 0, 0, 0);
#endif
            }
        };
        float sm[3][2]; // the finished sums of this half's rows
        if (kh == 0) {
            run(std::integral_constant<int, 2>{});
            *xrz = (f32x4v){acc[0][2], acc[0][3], acc[1][2], acc[1][3]};
            __syncthreads(); // X2
            const f32x2v nn = *xn;
            sm[0][0] = acc[0][0], sm[0][1] = acc[0][1], sm[1][0] = acc[1][0], sm[1][1] = acc[1][1], sm[2][0] = nn[0], sm[2][1] = nn[1];
        } else {
            run(std::integral_constant<int, 1>{});
            *xn = (f32x2v){acc[0][0], acc[0][1]};
            __syncthreads(); // X2
            const f32x4v rz = *xrz;
            sm[0][0] = rz[0], sm[0][1] = rz[1], sm[1][0] = rz[2], sm[1][1] = rz[3], sm[2][0] = acc[0][2], sm[2][1] = acc[0][3];
        }

        X4_T(1);
        char *nimg = lds + (cur ^ 1) * 2 * XF_IMG;
        unsigned pk[2]; // fp16 hi | lo << 16 of this lane's two new states
        float sv_r[2], sv_z[2], sv_n[2], sv_g[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float r = tt_fast_sigmoid(giv[0][e] + sm[0][e] * down);
            const float z = tt_fast_sigmoid(giv[1][e] + sm[1][e] * down);
            const float ghn = sm[2][e] * down;
            const float n = tt_fast_tanh(giv[2][e] + r * ghn);
            const float hn = (hreg[e] - n) * z + n;
            if (act[e])
                hreg[e] = hn;
            sv_r[e] = r;
            sv_z[e] = z;
            sv_n[e] = n;
            sv_g[e] = ghn;
            const float hs = hreg[e] * (float)(1 << X4_H_SHIFT);
            const _Float16 hi = (_Float16)hs;
            const _Float16 lo = (_Float16)(hs - (float)hi);
            _Float16 *dst = (_Float16 *)(nimg + (kq * 4 + 2 * kh + e) * XF_ROWB + own_off);
            dst[0] = hi;
            dst[XF_IMG / 2] = lo;
            pk[e] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
        // the next step's projections take their place HERE, in front of the sweep: behind a loop of loads the compiler no longer
        // knows what is in flight and waits with vmcnt(0) -- at the end of the step that meant waiting for the ten stores below
        if (s + 1 < steps) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    giv[g][e] = gnx[g][e];
        }
        X4_T(2);
#ifdef TT_X4_DBG
        tm[3] = tm[4] = tm[2];
#endif
        if (s + 1 < steps) {
            const unsigned tag = (unsigned)s + 1u;
            const int par = s & 1;
            // ---- publish: the even lane row 2 kh of units (2 q, 2 q + 1), the odd lane row 2 kh + 1 ----
            const unsigned g0 = swap1(odd ? pk[0] : pk[1]);
            const u32x4 v0 = odd ? (u32x4){g0, tag, pk[1], tag} : (u32x4){pk[0], tag, g0, tag};
            const int sbase = (par * 4 + m) * X4_REGION + send_off;
            if (same_xcd) // the partners read this XCD's L2
                __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, sbase, 0, 0);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, sbase, 0, 16); // aux 16 = sc1 (write-through)
            X4_T(3);
            // ---- sweep the other three members' granules (chunk tid of each) until every tag is this step's ----
            u32x4 got[3];
            unsigned spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const int om = (m + 1 + o) & 3;
                    got[o] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, (par * 4 + om) * X4_REGION + tid * 16, 0, 16);
                    ok = ok && got[o].y == tag && got[o].w == tag;
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
                sweep_backoff(spins);
            }
            X4_T(4);
            // chunk tid of a member's region: row urow, units 2 upr, 2 upr + 1 of that member (xf_chunk); a wave's 64 stores fall
            // on 16 rows x 4 dwords of one plane: conflict-free
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int om = (m + 1 + o) & 3;
                unsigned *dst = (unsigned *)(nimg + unpack_off + 32 * om);
                const unsigned a = got[o].x, b = got[o].z;
                dst[0] = (a & 0xffffu) | (b << 16);
                dst[XF_IMG / 4] = (a >> 16) | (b & 0xffff0000u);
            }
        }
        // (the step's bulk stores go out behind the hand-off: vector memory operations complete in issue order)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (act[e]) {
                if (d.out_seq)
                    d.out_seq[tok[e] * p.out_ld + d.out_col0 + unit] = hreg[e];
                if (d.gates) {
                    float *gs = d.gates + tok[e] * 4 * H + unit;
                    gs[0] = sv_r[e];
                    gs[H] = sv_z[e];
                    gs[2 * H] = sv_n[e];
                    gs[3 * H] = sv_g[e];
                }
            }
        X4_T(5);
        __syncthreads(); // X3: the next image is complete
        X4_T(6);
#ifdef TT_X4_DBG
        for (int i = 0; i < 6; ++i)
            tacc[i] += tm[i + 1] - tm[i];
#endif
        if (abort_flag)
            break;
        cur ^= 1;
    }
#ifdef TT_X4_DBG
    if (team == 0 && m == 1 && w == 0 && lane == 0 && steps > 40) {
        for (int i = 0; i < 6; ++i)
            atomicAdd(&x4_dbg[8 * kh + i], tacc[i]);
        atomicAdd(&x4_dbg[8 * kh + 6], (unsigned long long)steps);
    }
#endif
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (rid_e[e] >= 0)
            d.h_final[(size_t)rid_e[e] * H + unit] = hreg[e];
}

// ------------------------------------------------------------------ reverse-time recurrence on four CUs (training)
// dh_{t-1} = dh_t z + dGh W_hh, dGh = [dr_pre, dz_pre, dn_pre r] (16 rows x 3H).  Splitting the OUTPUT columns (as the forward
// kernel does) would need all 768 columns of dGh on every member: 72 KB of granules to receive per member and step.  The gate
// derivatives of a unit only need that unit's dh, so the team splits the REDUCTION instead: member m owns units
// [64 m, 64 m + 64), forms dGh for them (192 of the 768 columns, lane-local from the stash), multiplies that slice by ITS
// 192 ROWS of W_hh (fp16 hi/lo, 196 KB, resident in VGPRs) into a partial dh for ALL 256 units, keeps the 64 columns that
// are its own and publishes the other three 16 x 64 fp32 blocks to their owners as {value, tag} granules: 24 KB out and
// 24 KB in per member and step -- the forward kernel's volume -- straight from and into registers (no LDS staging).
// dh_own = dh z + (partial of member 0 + 1 + 2 + 3, always in that order): deterministic, and every partial carries its own
// per-row power-of-two scale (the row's largest |dGh| among the member's 192 columns), so the fp16 hi/lo split is at least as
// tight as in gru_bwd16_kernel, which scales a row by its maximum over all 768.  NOT bit-identical to that kernel (four
// partial chains of 6 k-steps instead of one of 24): checked against the oracle at the tests' gradient tolerance instead.
constexpr int XB_LDG = 192 + 8;            // fp16 elements per row of a dGh image (own 3 x 64 columns)
constexpr int XB_IMG = 16 * XB_LDG * 2;    // bytes of one (hi or lo) image
constexpr int XB_RM = 2 * 16 * 4 * 4;      // row maxima [buffer][row][wave], double-buffered
[[maybe_unused]] constexpr int XB_LDS = 2 * XB_IMG + XB_RM; // + 16 for the abort word (the four-wave member of the comparison build)
constexpr int XB_REGION = 16 * 64 * 8;     // one (dest, src) block of one parity
constexpr size_t XB_TEAM_BYTES = 2 * 4 * 4 * (size_t)XB_REGION; // [parity][dest][src]

struct GruSplitBwdParams {
    GruBwdParams g;
    char *xch;                  // [dir][team] x XB_TEAM_BYTES, zeroed before the launch
    int32_t *status;            // nullable: the backward call's own status word, bit 2 (value 4) on a time-out
    int nteams;
    unsigned spin_max;
};

#ifdef TT_AB // the four-wave member: superseded by gru_bwd16x4p_kernel, kept in the comparison build as its bit-identity reference
__global__ __launch_bounds__(256, 1) void gru_bwd16x4_kernel(GruSplitBwdParams sp)
{
    // One wave per SIMD: every instruction of the step is on the critical path (nothing else issues while it does), so the
    // step is written for instruction count -- 32-bit element offsets from uniform base pointers, unconditional loads of a
    // valid address + selects instead of exec-masked blocks, member indices RELATIVE to this member (compile-time register
    // indices), one exec-masked block of stores per row.
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int &abort_flag = *(int *)(lds + XB_LDS);
    const GruBwdParams &p = sp.g;
    const uint64_t drop_seed_v = (p.drop_p > 0.0f && p.drop_seed_ptr) ? *p.drop_seed_ptr : p.drop_seed;
    const int chunk = blockIdx.x >> 5, r32 = blockIdx.x & 31;
    const int m = __builtin_amdgcn_readfirstlane(r32 >> 3), team = chunk * 8 + (r32 & 7);
    if (team >= sp.nteams)
        return;
    const GruBwdDir d = p.dir[blockIdx.y];
    constexpr int H = X4_H, H3 = 3 * X4_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = team * ENC_RB;
    const int ul = 16 * w + j;          // this lane's unit within the member
    const int unit = 64 * m + ul;

    int len_e[4], off_e[4], rid_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int br = row0 + kq * 4 + e;
        rid_e[e] = br < p.B ? p.perm[br] : -1;
        len_e[e] = rid_e[e] >= 0 ? p.len[rid_e[e]] : 0;
        off_e[e] = rid_e[e] >= 0 ? p.tok_off[rid_e[e]] : 0;
    }
    int steps = max(max(len_e[0], len_e[1]), max(len_e[2], len_e[3]));
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));
    float dh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        dh[e] = (d.d_hfin && rid_e[e] >= 0) ? d.d_hfin[(size_t)rid_e[e] * H + unit] : 0.0f;
    const int exw = tt_pow2_exponent(*d.wmax);

    char *const img = lds;                            // [hi, lo][16][XB_LDG] fp16: this member's dGh columns [g][64]
    float *const rmax = (float *)(lds + 2 * XB_IMG);  // [2][16 rows][4 waves]
    if (tid == 0)
        abort_flag = 0;

    // ---- this wave's 48 fragments of W_hh: rows = the member's 192 gate rows (k-steps 8 g + 2 m + {0, 1} of gru16_pack_t's
    // order), columns = the 16 units [64 o + 16 w, + 16) of member o = (m + oo) & 3, oo = 0 (own) .. 3.  Packed order: wave
    // pw of 32 output units, fragment f = 4 s + 2 part + t: (pw, t) = (2 o + (w >> 1), w & 1) ----
    h8 wreg[4][6][2]; // [oo][k-step of the member's 192 rows][hi, lo]
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
        const int o = (m + oo) & 3;
        const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((const char *)d.wtp + (size_t)(2 * o + (w >> 1)) * 96 * 1024), 0, 96 * 1024, 0x00020000);
        const int loff = lane * 16 + (w & 1) * 1024;
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            const int sg = 8 * (s2 >> 1) + 2 * m + (s2 & 1);
#pragma unroll
            for (int part = 0; part < 2; ++part)
                wreg[oo][s2][part] = frag_load(wsrc, loff + (4 * sg + 2 * part) * 1024, 0);
        }
    }

    char *const xteam = sp.xch + ((size_t)blockIdx.y * sp.nteams + team) * (XB_TEAM_BYTES + X4_HEADER);
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xteam, 0, (int)XB_TEAM_BYTES + X4_HEADER, 0x00020000);
    // granule (row, col) of a block sits at (row 64 + col) 8; lanes j, j ^ 1 pair up: the even lane moves rows e = 0, 1 of
    // columns (ul, ul + 1), the odd lane rows e = 2, 3 of (ul - 1, ul), 16 bytes at a time
    const bool odd = j & 1;
    const int pair_off = ((kq * 4 + (odd ? 2 : 0)) * 64 + (ul & ~1)) * 8; // + 512 for the second row

    const int same_xcd = team_shares_xcd(xsrc, (int)XB_TEAM_BYTES, m, tid, sp.spin_max, &abort_flag + 1);
    if (same_xcd < 0) {
        if (tid == 0) {
            abort_flag = 1; // (poisons the bias sums below)
            int32_t *stw = sp.status;
            if (stw)
                atomicOr(stw, 4);
        }
        steps = 0;
    }

    // token of row e at step s: off + (s or len - 1 - s) while the row is live, the row's first token otherwise (a valid
    // address: what is loaded there is discarded by a select).  Everything below indexes with 32-bit element offsets from
    // the uniform base pointers (the host takes this kernel only when the largest array is below 4 GB).
    const unsigned ld = (unsigned)p.ld, cu = (unsigned)d.col0 + (unsigned)unit;
    struct Stash {
        float r[4], z[4], n[4], ghn[4], hp[4], dsv[4];
    };
    auto load_stash = [&](int s, Stash &st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = s >= 0 && s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const unsigned tok = (unsigned)(off_e[e] + (a ? t : 0));
            const unsigned ptok = (a && s > 0) ? (d.reverse ? tok + 1u : tok - 1u) : tok;
            const unsigned go = tok * (4u * H) + (unsigned)unit;
            st.r[e] = d.gates[go];
            st.z[e] = d.gates[go + H];
            st.n[e] = d.gates[go + 2 * H];
            st.ghn[e] = d.gates[go + 3 * H];
            const float hpv = d.hseq[ptok * ld + cu];
            st.hp[e] = s > 0 ? hpv : 0.0f;
            st.dsv[e] = d.d_seq ? d.d_seq[tok * ld + cu] : 0.0f;
        }
    };
    Stash cur_st, next_st;
    load_stash(steps - 1, cur_st);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): fragments and the first stash are in (no such waits inside the loop)
    __syncthreads();
    int rb = 0;
    float bsum[4] = {0, 0, 0, 0}; // column sums over this lane's rows and all steps: dr, dz, dn, dn r
    float mx_i = 0.0f, mx_h = 0.0f;
    unsigned it = 0;              // steps done: tag = it + 1, parity = it & 1

    for (int s = steps - 1; s >= 0; --s, ++it) {
        float direct[4], gv[3][4], dnp[4], mrow[4];
        bool act[4];
        unsigned tokv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tokv[e] = (unsigned)(off_e[e] + (act[e] ? t : 0));
            const float r = cur_st.r[e], z = cur_st.z[e], n = cur_st.n[e], ghn = cur_st.ghn[e], hp = cur_st.hp[e];
            float dsv = cur_st.dsv[e];
            if (d.d_seq && p.drop_p > 0.0f) // (uniform)
                dsv *= tt_dropout_scale(drop_seed_v, p.drop_layer, ((uint64_t)(rid_e[e] < 0 ? 0 : rid_e[e]) * p.T + t) * p.ld + cu,
                                        p.drop_p);
            const float dhv = dh[e] + dsv;
            const float dn_pre = dhv * (1.0f - z) * (1.0f - n * n);
            const float dz_pre = dhv * (hp - n) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn_v = dn_pre * r;
            gv[0][e] = act[e] ? dr_pre : 0.0f;
            gv[1][e] = act[e] ? dz_pre : 0.0f;
            gv[2][e] = act[e] ? dghn_v : 0.0f;
            dnp[e] = act[e] ? dn_pre : 0.0f;
            direct[e] = dhv * z;
            bsum[0] += gv[0][e];
            bsum[1] += gv[1][e];
            bsum[2] += dnp[e];
            bsum[3] += gv[2][e];
            const float m2 = fmaxf(fabsf(gv[0][e]), fabsf(gv[1][e]));
            mx_i = fmaxf(mx_i, fmaxf(m2, fabsf(dnp[e])));
            mrow[e] = fmaxf(m2, fabsf(gv[2][e]));
            mx_h = fmaxf(mx_h, mrow[e]);
        }
        // row maxima over the member's 192 columns: 16 lanes of a kq group -> one LDS word per wave and row
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mrow[e] = fmaxf(mrow[e], row_ror<1>(mrow[e])); // after rotations by 1, 2, 4, 8 every lane of the row holds its maximum
            mrow[e] = fmaxf(mrow[e], row_ror<2>(mrow[e]));
            mrow[e] = fmaxf(mrow[e], row_ror<4>(mrow[e]));
            mrow[e] = fmaxf(mrow[e], row_ror<8>(mrow[e]));
            if (j == 0)
                rmax[(rb * 16 + kq * 4 + e) * 4 + w] = mrow[e];
        }
        load_stash(s - 1, next_st); // in flight during the MFMAs and the hand-off below
        __syncthreads();            // B1: row maxima visible; every wave is done reading the previous step's images
        if (abort_flag) // (set, if at all, before its wave reached B1: every wave reads the same value here)
            break;
        float down[4]; // rows 4 kq + e: the rows of this lane's values AND of its accumulators (the MFMA's C layout)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4v rm = *(const f32x4v *)(rmax + (rb * 16 + kq * 4 + e) * 4); // the four waves' maxima of this row
            const float mm = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
            const int er = tt_pow2_exponent(__float_as_uint(mm));
            const float upr = ldexpf(1.0f, er);
            down[e] = ldexpf(1.0f, -(er + exw));
            _Float16 *dst = (_Float16 *)img + (kq * 4 + e) * XB_LDG + ul;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float x = gv[g][e] * upr;
                const _Float16 hi = (_Float16)x;
                dst[g * 64] = hi;
                dst[XB_IMG / 2 + g * 64] = (_Float16)(x - (float)hi);
            }
        }
        rb ^= 1;
        __syncthreads(); // B2: the dGh images are complete

        f32x4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}; // [oo]: destination member (m + oo) & 3
        const char *arow = img + j * (XB_LDG * 2) + kq * 16;
        h8 a_hi[2], a_lo[2];
        a_hi[0] = *(const h8 *)(arow);
        a_lo[0] = *(const h8 *)(arow + XB_IMG);
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            if (s2 + 1 < 6) {
                a_hi[(s2 + 1) & 1] = *(const h8 *)(arow + (s2 + 1) * 64);
                a_lo[(s2 + 1) & 1] = *(const h8 *)(arow + XB_IMG + (s2 + 1) * 64);
            }
#pragma unroll
            for (int oo = 0; oo < 4; ++oo)
                acc[oo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[oo][s2][0], acc[oo], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 2)
#pragma unroll
            for (int oo = 0; oo < 4; ++oo)
                acc[oo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], wreg[oo][s2][0], acc[oo], 0, 0, 0);
#pragma unroll
            for (int oo = 0; oo < 4; ++oo)
                acc[oo] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[oo][s2][1], acc[oo], 0, 0, 0);
#endif
        }
        // partial dh of this member: part[oo][e] = rows 4 kq + e, unit 16 w + j of member (m + oo) & 3
        float part[4][4];
#pragma unroll
        for (int oo = 0; oo < 4; ++oo)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                part[oo][e] = acc[oo][e] * down[e];
        float sum[4] = {0, 0, 0, 0};
        if (s > 0) { // (the partials of the last step would only feed a dh nobody reads)
            const unsigned tag = it + 1u;
            const int par = (int)(it & 1u);
            // ---- publish the three foreign blocks: [parity][dest][src = m] ----
#pragma unroll
            for (int oo = 1; oo < 4; ++oo) {
                const int o = (m + oo) & 3;
                const unsigned p0 = __float_as_uint(part[oo][0]), p1 = __float_as_uint(part[oo][1]);
                const unsigned p2 = __float_as_uint(part[oo][2]), p3 = __float_as_uint(part[oo][3]);
                const unsigned g0 = swap1(odd ? p0 : p2), g1 = swap1(odd ? p1 : p3);
                const u32x4 v0 = odd ? (u32x4){g0, tag, p2, tag} : (u32x4){p0, tag, g0, tag};
                const u32x4 v1 = odd ? (u32x4){g1, tag, p3, tag} : (u32x4){p1, tag, g1, tag};
                const int base = ((par * 4 + o) * 4 + m) * XB_REGION + pair_off;
                if (same_xcd) { // the partners read this XCD's L2
                    __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, base, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, base + 512, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, base, 0, 16); // aux 16 = sc1 (write-through)
                    __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, base + 512, 0, 16);
                }
            }
            // ---- sweep the three blocks addressed to this member until every tag is this step's ----
            u32x4 got[3][2];
            unsigned spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int oo = 1; oo < 4; ++oo) {
                    const int src = (m + oo) & 3;
                    const int base = ((par * 4 + m) * 4 + src) * XB_REGION + pair_off;
                    got[oo - 1][0] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, base, 0, 16);
                    got[oo - 1][1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, base + 512, 0, 16);
                    ok = ok && got[oo - 1][0].y == tag && got[oo - 1][0].w == tag && got[oo - 1][1].y == tag && got[oo - 1][1].w == tag;
                }
                if (__all(ok))
                    break;
                if (++spins > sp.spin_max) {
                    if (lane == 0) {
                        abort_flag = 1;
                        int32_t *stw = sp.status;
                        if (stw)
                            atomicOr(stw, 4);
                    }
                    break;
                }
                sweep_backoff(spins);
            }
            // un-pair: this lane's column is the even (x) or the odd (z) word; the neighbour holds its other two rows
            float th[4][4]; // [oo][e]: the partial of member (m + oo) & 3 for this lane's unit
#pragma unroll
            for (int e = 0; e < 4; ++e)
                th[0][e] = part[0][e];
#pragma unroll
            for (int oo = 1; oo < 4; ++oo) {
                const u32x4 a0 = got[oo - 1][0], a1 = got[oo - 1][1];
                const unsigned mine0 = odd ? a0.z : a0.x, mine1 = odd ? a1.z : a1.x; // rows (2, 3) | (0, 1) of my column
                const unsigned h0 = swap1(odd ? a0.x : a0.z), h1 = swap1(odd ? a1.x : a1.z); // my column, the neighbour's rows
                th[oo][0] = __uint_as_float(odd ? h0 : mine0);
                th[oo][1] = __uint_as_float(odd ? h1 : mine1);
                th[oo][2] = __uint_as_float(odd ? mine0 : h0);
                th[oo][3] = __uint_as_float(odd ? mine1 : h1);
            }
            // members 0, 1, 2, 3 -- always in that order, whoever this member is: oo = (q - m) & 3 of absolute member q
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t;
                switch (m) { // (uniform)
                case 0: t = ((th[0][e] + th[1][e]) + th[2][e]) + th[3][e]; break;
                case 1: t = ((th[3][e] + th[0][e]) + th[1][e]) + th[2][e]; break;
                case 2: t = ((th[2][e] + th[3][e]) + th[0][e]) + th[1][e]; break;
                default: t = ((th[1][e] + th[2][e]) + th[3][e]) + th[0][e]; break;
                }
                sum[e] = t;
            }
        }
        // the step's gradient rows go out BEHIND the hand-off (in front of it, 24 stores per lane would complete before the
        // sweep's loads could: vector memory operations finish in issue order); they drain under the next step's work
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (act[e]) {
                const unsigned go = tokv[e] * (unsigned)H3 + (unsigned)unit;
                d.dgi[go] = gv[0][e];
                d.dgi[go + H] = gv[1][e];
                d.dgi[go + 2 * H] = dnp[e];
                d.dghn[go] = gv[0][e];
                d.dghn[go + H] = gv[1][e];
                d.dghn[go + 2 * H] = gv[2][e];
                dh[e] = direct[e] + sum[e];
            }
        cur_st = next_st;
    }
    __syncthreads();
    // bias gradients of this member's units for the row group, and the operand maxima
    if (d.bias_slab) {
        float *slab = d.bias_slab + (size_t)team * 2 * H3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = bsum[g];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            bsum[g] = abort_flag ? __uint_as_float(0x7fc00000u) : v; // a team that gave up poisons its sums: NaN gradients, not wrong ones
        }
        if (kq == 0) {
            slab[unit] = bsum[0];
            slab[H + unit] = bsum[1];
            slab[2 * H + unit] = bsum[2];
            slab[H3 + unit] = bsum[0];
            slab[H3 + H + unit] = bsum[1];
            slab[H3 + 2 * H + unit] = bsum[3];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mx_i = fmaxf(mx_i, __shfl_xor(mx_i, off));
            mx_h = fmaxf(mx_h, __shfl_xor(mx_h, off));
        }
        if (lane == 0) {
            atomicMax(d.mx_dgi, __float_as_uint(mx_i));
            atomicMax(d.mx_dghn, __float_as_uint(mx_h));
        }
    }
}
#endif // TT_AB

// ------------------------------------------------------------------ the same recurrence with TWO waves per SIMD
// A lone wave issues one instruction every ~8.7 clocks whatever its kind; two waves on a SIMD each do (tools/experiments/
// valu_rate.hip) -- and gru_bwd16x4_kernel's step is nothing but its instruction count (~4 ns per instruction of the step).
// Here a member is EIGHT waves: the pair (w, w + 4) shares unit slice [16 w, 16 w + 16) and halves the step between its waves
// (half kh = wave >> 2):
//   * gate derivatives, row maxima, image writes, stash loads and gradient stores of rows 4 kq + 2 kh + {0, 1} (two of the
//     lane's four rows);
//   * the products for destination members oo = 2 kh, 2 kh + 1 (36 MFMAs, 24 resident W fragments = 96 registers): every
//     accumulator is still one chain over the member's six k-steps, so partials, sums and dh are BIT-IDENTICAL to
//     gru_bwd16x4_kernel's and the granule traffic is the same (half 0 publishes one block, half 1 two);
//   * the receive side for its two rows: ONE 16-byte load per source member (rows 2 kh | 2 kh + 1 on the even | odd lane of a
//     pair); half 1 gets the member's own partial (oo = 0, computed by half 0) through LDS -- a third barrier per step.
// The row scale factors are computed once per row (by the lanes that own the row) and read back as one LDS vector per lane.
// Bias sums accumulate per half and are added at the end (half 0 + half 1): deterministic, not bit-identical to the
// four-wave kernel's interleaved order.
// The dGh image (fp16 hi / lo) is laid out as gru_seq16x4p_kernel's h image, for the same reason: [kq plane][row][k-step] x 16
// bytes, plane stride a multiple of 256 bytes, 7 slots per row -- the A-fragment reads are conflict-free (row-major with
// 400-byte rows: one 2-way conflict in each of a read's four lane groups, SQ_LDS_BANK_CONFLICT 33 % of SQ_LDS_IDX_ACTIVE).
constexpr int XP_ROWB = 6 * 16 + 16;             // bytes of one row in one plane: 6 k-steps x 16 B + one slot of padding
constexpr int XP_PLANE = 16 * XP_ROWB;           // 1792 = 7 x 256
constexpr int XP_IMG = 4 * XP_PLANE;             // bytes of one (hi or lo) image
constexpr int XP_BASE = 2 * XP_IMG + XB_RM;      // images + row maxima
constexpr int XP_RDOWN = 2 * 16 * 4;             // [buffer][row] 2^-(e_row + e_W)
constexpr int XP_PART = 4 * 64 * 8;              // [unit slice w][lane] rows 2, 3 of the member's own partial
constexpr int XP_BIAS = 4 * 64 * 16;             // [w][lane] half 1's four bias sums (end of the kernel)
constexpr int XP_LDS = XP_BASE + XP_RDOWN + XP_PART + XP_BIAS; // + 16 for the abort word

#ifdef TT_X4_DBG
__device__ unsigned long long xb_dbg[32];
#endif
__global__ __launch_bounds__(512, 1) void gru_bwd16x4p_kernel(GruSplitBwdParams sp)
{
#ifdef TT_X4_DBG
    unsigned long long tm[10], tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int &abort_flag = *(int *)(lds + XP_LDS);
    const GruBwdParams &p = sp.g;
    const uint64_t drop_seed_v = (p.drop_p > 0.0f && p.drop_seed_ptr) ? *p.drop_seed_ptr : p.drop_seed;
    const int chunk = blockIdx.x >> 5, r32 = blockIdx.x & 31;
    const int m = __builtin_amdgcn_readfirstlane(r32 >> 3), team = chunk * 8 + (r32 & 7);
    if (team >= sp.nteams)
        return;
    const GruBwdDir d = p.dir[blockIdx.y];
    constexpr int H = X4_H, H3 = 3 * X4_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3, kh = wv >> 2;
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = team * ENC_RB;
    const int ul = 16 * w + j;          // this lane's unit within the member
    const int unit = 64 * m + ul;

    int len_e[2], off_e[2], rid_e[2], steps = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { // (all four rows of the lane for the step count, two of them kept)
        const int br = row0 + kq * 4 + e;
        const int rid = br < p.B ? p.perm[br] : -1;
        const int len = rid >= 0 ? p.len[rid] : 0;
        steps = max(steps, len);
        if ((e >> 1) == kh) {
            rid_e[e & 1] = rid;
            len_e[e & 1] = len;
            off_e[e & 1] = rid >= 0 ? p.tok_off[rid] : 0;
        }
    }
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));
    float dh[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        dh[e] = (d.d_hfin && rid_e[e] >= 0) ? d.d_hfin[(size_t)rid_e[e] * H + unit] : 0.0f;
    const int exw = tt_pow2_exponent(*d.wmax);

    static_assert(XP_PLANE % 256 == 0 && (XP_ROWB / 16) % 2 == 1, "plane stride a multiple of the bank row, row stride an odd number of 16-B slots");
    char *const img = lds;                                       // [hi, lo][kq plane 4][row 16][k-step 6] x 16 B (fp16)
    float *const rmax = (float *)(lds + 2 * XP_IMG);             // [2][16 rows][4 unit slices]
    float *const rdown = (float *)(lds + XP_BASE);               // [2][16 rows]
    float *const xpart = (float *)(lds + XP_BASE + XP_RDOWN);    // [4][64 lanes][2]
    float *const xbias = (float *)(lds + XP_BASE + XP_RDOWN + XP_PART); // [4][64 lanes][4]
    // where this lane's column of gate 0 sits in an image row (bytes): column c = 64 g + 16 w + j -> k-step c >> 5 = 2 g + (w >> 1),
    // fragment lane group (c & 31) >> 3 = 2 (w & 1) + (j >> 3), element j & 7; gate g adds two k-steps
    const int own_off = (2 * (w & 1) + (j >> 3)) * XP_PLANE + (w >> 1) * 16 + (j & 7) * 2;
    if (tid == 0)
        abort_flag = 0;

    // ---- this wave's 24 fragments of W_hh: destinations oo = 2 kh + o2 (gru_bwd16x4_kernel's order and addressing) ----
    h8 wreg[2][6][2];
#pragma unroll
    for (int o2 = 0; o2 < 2; ++o2) {
        const int o = (m + 2 * kh + o2) & 3;
        const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((const char *)d.wtp + (size_t)(2 * o + (w >> 1)) * 96 * 1024), 0, 96 * 1024, 0x00020000);
        const int loff = lane * 16 + (w & 1) * 1024;
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            const int sg = 8 * (s2 >> 1) + 2 * m + (s2 & 1);
#pragma unroll
            for (int part = 0; part < 2; ++part)
                wreg[o2][s2][part] = frag_load(wsrc, loff + (4 * sg + 2 * part) * 1024, 0);
        }
    }

    char *const xteam = sp.xch + ((size_t)blockIdx.y * sp.nteams + team) * (XB_TEAM_BYTES + X4_HEADER);
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xteam, 0, (int)XB_TEAM_BYTES + X4_HEADER, 0x00020000);
    const bool odd = j & 1;
    // send (all four rows of a block, as the four-wave kernel): the even lane moves rows 0, 1 of columns (ul, ul + 1), the odd
    // lane rows 2, 3 of (ul - 1, ul); receive (this half's two rows): the even lane row 2 kh, the odd lane row 2 kh + 1
    const int send_off = ((kq * 4 + (odd ? 2 : 0)) * 64 + (ul & ~1)) * 8; // + 512 for the second row
    const int recv_off = ((kq * 4 + 2 * kh + (odd ? 1 : 0)) * 64 + (ul & ~1)) * 8;

    const int same_xcd = team_shares_xcd(xsrc, (int)XB_TEAM_BYTES, m, tid, sp.spin_max, &abort_flag + 1);
    if (same_xcd < 0) {
        if (tid == 0) {
            abort_flag = 1; // (poisons the bias sums below)
            int32_t *stw = sp.status;
            if (stw)
                atomicOr(stw, 4);
        }
        steps = 0;
    }

    const unsigned ld = (unsigned)p.ld, cu = (unsigned)d.col0 + (unsigned)unit;
    struct Stash {
        float r[2], z[2], n[2], ghn[2], hp[2], dsv[2];
    };
    auto load_stash = [&](int s, Stash &st) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool a = s >= 0 && s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const unsigned tok = (unsigned)(off_e[e] + (a ? t : 0));
            const unsigned ptok = (a && s > 0) ? (d.reverse ? tok + 1u : tok - 1u) : tok;
            const unsigned go = tok * (4u * H) + (unsigned)unit;
            st.r[e] = d.gates[go];
            st.z[e] = d.gates[go + H];
            st.n[e] = d.gates[go + 2 * H];
            st.ghn[e] = d.gates[go + 3 * H];
            const float hpv = d.hseq[ptok * ld + cu];
            st.hp[e] = s > 0 ? hpv : 0.0f;
            st.dsv[e] = d.d_seq ? d.d_seq[tok * ld + cu] : 0.0f;
        }
    };
    Stash cur_st, next_st;
    load_stash(steps - 1, cur_st);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): fragments and the first stash are in (no such waits inside the loop)
    __syncthreads();
    int rb = 0;
    float bsum[4] = {0, 0, 0, 0}; // column sums over this lane's rows and all steps: dr, dz, dn, dn r
    float mx_i = 0.0f, mx_h = 0.0f;
    unsigned it = 0;              // steps done: tag = it + 1, parity = it & 1

    for (int s = steps - 1; s >= 0; --s, ++it) {
        X4_T(0);
        float direct[2], gv[3][2], dnp[2], mrow[2];
        bool act[2];
        unsigned tokv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tokv[e] = (unsigned)(off_e[e] + (act[e] ? t : 0));
            const float r = cur_st.r[e], z = cur_st.z[e], n = cur_st.n[e], ghn = cur_st.ghn[e], hp = cur_st.hp[e];
            float dsv = cur_st.dsv[e];
            if (d.d_seq && p.drop_p > 0.0f) // (uniform)
                dsv *= tt_dropout_scale(drop_seed_v, p.drop_layer, ((uint64_t)(rid_e[e] < 0 ? 0 : rid_e[e]) * p.T + t) * p.ld + cu,
                                        p.drop_p);
            const float dhv = dh[e] + dsv;
            const float dn_pre = dhv * (1.0f - z) * (1.0f - n * n);
            const float dz_pre = dhv * (hp - n) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            const float dghn_v = dn_pre * r;
            gv[0][e] = act[e] ? dr_pre : 0.0f;
            gv[1][e] = act[e] ? dz_pre : 0.0f;
            gv[2][e] = act[e] ? dghn_v : 0.0f;
            dnp[e] = act[e] ? dn_pre : 0.0f;
            direct[e] = dhv * z;
            bsum[0] += gv[0][e];
            bsum[1] += gv[1][e];
            bsum[2] += dnp[e];
            bsum[3] += gv[2][e];
            const float m2 = fmaxf(fabsf(gv[0][e]), fabsf(gv[1][e]));
            mx_i = fmaxf(mx_i, fmaxf(m2, fabsf(dnp[e])));
            mrow[e] = fmaxf(m2, fabsf(gv[2][e]));
            mx_h = fmaxf(mx_h, mrow[e]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            mrow[e] = fmaxf(mrow[e], row_ror<1>(mrow[e]));
            mrow[e] = fmaxf(mrow[e], row_ror<2>(mrow[e]));
            mrow[e] = fmaxf(mrow[e], row_ror<4>(mrow[e]));
            mrow[e] = fmaxf(mrow[e], row_ror<8>(mrow[e]));
            if (j == 0)
                rmax[(rb * 16 + kq * 4 + 2 * kh + e) * 4 + w] = mrow[e];
        }
        X4_T(1);
        load_stash(s - 1, next_st); // in flight during the MFMAs and the hand-off below
        X4_T(2);
        __syncthreads();            // B1: row maxima visible; every wave is done reading the previous step's images
        X4_T(3);
        if (abort_flag) // (set, if at all, before its wave reached B1: every wave reads the same value here)
            break;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = kq * 4 + 2 * kh + e;
            const f32x4v rm = *(const f32x4v *)(rmax + (rb * 16 + row) * 4); // the four unit slices' maxima of this row
            const float mm = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
            const int er = tt_pow2_exponent(__float_as_uint(mm));
            const float upr = ldexpf(1.0f, er);
            if (w == 0 && j == 0)
                rdown[rb * 16 + row] = ldexpf(1.0f, -(er + exw));
            _Float16 *dst = (_Float16 *)(img + row * XP_ROWB + own_off);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float x = gv[g][e] * upr;
                const _Float16 hi = (_Float16)x;
                dst[g * 16] = hi;                                      // (+ 32 bytes: two k-steps)
                dst[XP_IMG / 2 + g * 16] = (_Float16)(x - (float)hi);
            }
        }
        X4_T(4);
        __syncthreads(); // B2: the dGh images and the row factors are complete
        X4_T(5);
        const f32x4v down = *(const f32x4v *)(rdown + rb * 16 + kq * 4); // rows 4 kq + 0 .. 3: the rows of this lane's accumulators
        rb ^= 1;

        f32x4v acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // destination members (m + 2 kh + o2) & 3
        const char *arow = img + kq * XP_PLANE + j * XP_ROWB; // (row j, fragment lane group kq: conflict-free b128 reads, as gru_seq16x4p)
        h8 a_hi[2], a_lo[2];
        a_hi[0] = *(const h8 *)(arow);
        a_lo[0] = *(const h8 *)(arow + XP_IMG);
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            if (s2 + 1 < 6) {
                a_hi[(s2 + 1) & 1] = *(const h8 *)(arow + (s2 + 1) * 16);
                a_lo[(s2 + 1) & 1] = *(const h8 *)(arow + XP_IMG + (s2 + 1) * 16);
            }
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2)
                acc[o2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[o2][s2][0], acc[o2], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 2)
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2)
                acc[o2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], wreg[o2][s2][0], acc[o2], 0, 0, 0);
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2)
                acc[o2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], wreg[o2][s2][1], acc[o2], 0, 0, 0);
#endif
        }
        float part[2][4]; // [o2][row 4 kq + e of the accumulator]
#pragma unroll
        for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                part[o2][e] = acc[o2][e] * down[e];
        float sum[2] = {0, 0};
        // (the next step's stash takes its place here, in front of the sweep: behind a loop of loads the compiler waits with
        //  vmcnt(0), which at the end of the step would include the twelve gradient stores)
        cur_st = next_st;
        X4_T(6);
#ifdef TT_X4_DBG
        tm[7] = tm[8] = tm[6];
#endif
        if (s > 0) { // (the partials of the last step would only feed a dh nobody reads)
            const unsigned tag = it + 1u;
            const int par = (int)(it & 1u);
            // ---- publish the foreign blocks of this half: oo = 1 (half 0), oo = 2, 3 (half 1); [parity][dest][src = m] ----
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2) {
                if (kh == 0 && o2 == 0) // (uniform) the member's own block stays here
                    continue;
                const int o = (m + 2 * kh + o2) & 3;
                const unsigned p0 = __float_as_uint(part[o2][0]), p1 = __float_as_uint(part[o2][1]);
                const unsigned p2 = __float_as_uint(part[o2][2]), p3 = __float_as_uint(part[o2][3]);
                const unsigned g0 = swap1(odd ? p0 : p2), g1 = swap1(odd ? p1 : p3);
                const u32x4 v0 = odd ? (u32x4){g0, tag, p2, tag} : (u32x4){p0, tag, g0, tag};
                const u32x4 v1 = odd ? (u32x4){g1, tag, p3, tag} : (u32x4){p1, tag, g1, tag};
                const int base = ((par * 4 + o) * 4 + m) * XB_REGION + send_off;
                if (same_xcd) { // the partners read this XCD's L2
                    __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, base, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, base + 512, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(v0, xsrc, base, 0, 16); // aux 16 = sc1 (write-through)
                    __builtin_amdgcn_raw_buffer_store_b128(v1, xsrc, base + 512, 0, 16);
                }
            }
            if (kh == 0) // rows 2, 3 of the member's own partial for the other half
                *(__attribute__((ext_vector_type(2))) float *)(xpart + (w * 64 + lane) * 2) = (__attribute__((ext_vector_type(2))) float){part[0][2], part[0][3]};
            X4_T(7);
            // ---- sweep this half's rows of the three blocks addressed to this member until every tag is this step's ----
            u32x4 got[3];
            unsigned spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int oo = 1; oo < 4; ++oo) {
                    const int src = (m + oo) & 3;
                    got[oo - 1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, ((par * 4 + m) * 4 + src) * XB_REGION + recv_off, 0, 16);
                    ok = ok && got[oo - 1].y == tag && got[oo - 1].w == tag;
                }
                if (__all(ok))
                    break;
                if (++spins > sp.spin_max) {
                    if (lane == 0) {
                        abort_flag = 1;
                        int32_t *stw = sp.status;
                        if (stw)
                            atomicOr(stw, 4);
                    }
                    break;
                }
                sweep_backoff(spins);
            }
            X4_T(8);
            __syncthreads(); // B3: half 0's rows 2, 3 of the own partial are in LDS
            float th[4][2]; // [oo][e]: the partial of member (m + oo) & 3 for this lane's unit, rows 2 kh + e
            if (kh == 0) {
                th[0][0] = part[0][0];
                th[0][1] = part[0][1];
            } else {
                const __attribute__((ext_vector_type(2))) float own = *(const __attribute__((ext_vector_type(2))) float *)(xpart + (w * 64 + lane) * 2);
                th[0][0] = own.x;
                th[0][1] = own.y;
            }
#pragma unroll
            for (int oo = 1; oo < 4; ++oo) {
                const u32x4 a = got[oo - 1];
                // even lane: row 2 kh of columns (ul, ul + 1); odd lane: row 2 kh + 1 of (ul - 1, ul)
                const unsigned other = swap1(odd ? a.x : a.z); // my column, the neighbour's row
                th[oo][0] = __uint_as_float(odd ? other : a.x);
                th[oo][1] = __uint_as_float(odd ? a.z : other);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float t;
                switch (m) { // (uniform) members 0, 1, 2, 3 -- always in that order
                case 0: t = ((th[0][e] + th[1][e]) + th[2][e]) + th[3][e]; break;
                case 1: t = ((th[3][e] + th[0][e]) + th[1][e]) + th[2][e]; break;
                case 2: t = ((th[2][e] + th[3][e]) + th[0][e]) + th[1][e]; break;
                default: t = ((th[1][e] + th[2][e]) + th[3][e]) + th[0][e]; break;
                }
                sum[e] = t;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (act[e]) {
                const unsigned go = tokv[e] * (unsigned)H3 + (unsigned)unit;
                d.dgi[go] = gv[0][e];
                d.dgi[go + H] = gv[1][e];
                d.dgi[go + 2 * H] = dnp[e];
                d.dghn[go] = gv[0][e];
                d.dghn[go + H] = gv[1][e];
                d.dghn[go + 2 * H] = gv[2][e];
                dh[e] = direct[e] + sum[e];
            }
        X4_T(9);
#ifdef TT_X4_DBG
        for (int i = 0; i < 9; ++i)
            tacc[i] += tm[i + 1] - tm[i];
#endif
    }
#ifdef TT_X4_DBG
    if (team == 0 && m == 1 && w == 0 && lane == 0 && steps > 40) {
        for (int i = 0; i < 9; ++i)
            atomicAdd(&xb_dbg[16 * kh + i], tacc[i]);
        atomicAdd(&xb_dbg[16 * kh + 9], (unsigned long long)steps);
    }
#endif
    __syncthreads();
    if (d.bias_slab) {
        float *slab = d.bias_slab + (size_t)team * 2 * H3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = bsum[g];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            bsum[g] = v;
        }
        if (kh == 1)
            *(f32x4v *)(xbias + (w * 64 + lane) * 4) = (f32x4v){bsum[0], bsum[1], bsum[2], bsum[3]};
        __syncthreads();
        if (kh == 0 && kq == 0) {
            const f32x4v o = *(const f32x4v *)(xbias + (w * 64 + lane) * 4);
            const float nanv = __uint_as_float(0x7fc00000u); // a team that gave up poisons its sums: NaN gradients, not wrong ones
            const float b0 = abort_flag ? nanv : bsum[0] + o[0], b1 = abort_flag ? nanv : bsum[1] + o[1];
            const float b2 = abort_flag ? nanv : bsum[2] + o[2], b3 = abort_flag ? nanv : bsum[3] + o[3];
            slab[unit] = b0;
            slab[H + unit] = b1;
            slab[2 * H + unit] = b2;
            slab[H3 + unit] = b0;
            slab[H3 + H + unit] = b1;
            slab[H3 + 2 * H + unit] = b3;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mx_i = fmaxf(mx_i, __shfl_xor(mx_i, off));
            mx_h = fmaxf(mx_h, __shfl_xor(mx_h, off));
        }
        if (lane == 0) {
            atomicMax(d.mx_dgi, __float_as_uint(mx_i));
            atomicMax(d.mx_dghn, __float_as_uint(mx_h));
        }
    }
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

// The caller chooses between these kernels and the one-workgroup recurrences per call (TT_ENC_ONE_WORKGROUP, include/tt.h); the
// comparison build (-DTT_AB) can also switch them off from the environment (TT_GRU_SPLIT=0 / TT_GRU_SPLIT_BWD=0) and select
// the four-wave members (=4).
static bool split_enabled() { return TT_AB_SWITCH(TT_GRU_SPLIT, 1) != 0; }

size_t gru16x4_xch_bytes(int B, int H, int ndir)
{
    if (H != X4_H || B <= 0 || B > 1024) // (more than 64 row groups never fit one workgroup per CU four times over)
        return 0;
    return (size_t)ndir * ((B + ENC_RB - 1) / ENC_RB) * (X4_TEAM_BYTES + X4_HEADER);
}

// one workgroup per CU at most, so that every member of every team is resident (one wave per SIMD with the whole
// register file: nothing else fits on a CU beside one of these workgroups)
// Launches a call of this shape takes: 1 = both directions' teams in one grid, 2 = one launch per direction (a bidirectional
// layer whose two directions together ask for more CUs than the device has while each alone fits: 1 024 rows x 2 directions on
// 256 CUs -- the directions are independent recurrences, and one behind the other on four CUs per row group is faster than both
// at once on one CU per row group: 2 x 231 against 709 us forward, 2 x 353 against 1 023 us backward for the reference's default
// model shape at 512 triplets), 0 = the shape does not fit.
int gru16x4_launches(int B, int H, int ndir)
{
    if (!split_enabled() || H != X4_H || B <= 0)
        return 0;
    const int per_dir = ((B + ENC_RB - 1) / ENC_RB + 7) / 8 * 32;
    if (per_dir * ndir <= device_cus())
        return 1;
    return ndir == 2 && per_dir <= device_cus() ? 2 : 0;
}

bool gru16x4_usable(int B, int H, int ndir) { return gru16x4_launches(B, H, ndir) > 0; }

size_t gru16x4_bwd_xch_bytes(int B, int H, int ndir)
{
    if (H != X4_H || B <= 0 || B > 1024)
        return 0;
    return (size_t)ndir * ((B + ENC_RB - 1) / ENC_RB) * (XB_TEAM_BYTES + X4_HEADER);
}

bool gru16x4_bwd_usable(int B, int H, int ndir)
{
    return TT_AB_SWITCH(TT_GRU_SPLIT_BWD, 1) != 0 && gru16x4_usable(B, H, ndir);
}

int gru16x4_bwd_launch(const GruBwdParams &bp, int ndir, void *xch, int32_t *status, hipStream_t st)
{
    GruSplitBwdParams sp;
    sp.g = bp;
    sp.xch = (char *)xch;
    sp.status = status;
    sp.nteams = (bp.B + ENC_RB - 1) / ENC_RB;
    sp.spin_max = 1u << 17;
    TT_RC_CHECK(tt_zero_async(xch, gru16x4_bwd_xch_bytes(bp.B, bp.H, ndir), st));
    static bool attr_done = false;
    if (!attr_done) {
#ifdef TT_AB
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_bwd16x4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS + 16));
#endif
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_bwd16x4p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XP_LDS + 16));
        attr_done = true;
    }
#ifdef TT_X4_DBG
    static int calls = 0;
    if (++calls % 8 == 0) {
        unsigned long long h[32];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(xb_dbg), sizeof h) == hipSuccess && h[9] && h[25])
            for (int k = 0; k < 2; ++k) {
                const unsigned long long *q = h + 16 * k;
                fprintf(stderr, "xbdbg half %d: clocks per step: gate derivatives %.0f, stash requests %.0f, B1 %.0f, scale + image %.0f, B2 %.0f, matrix %.0f, publish %.0f, sweep %.0f, B3 + sum + stores %.0f\n", k,
                        (double)q[0] / q[9], (double)q[1] / q[9], (double)q[2] / q[9], (double)q[3] / q[9], (double)q[4] / q[9], (double)q[5] / q[9],
                        (double)q[6] / q[9], (double)q[7] / q[9], (double)q[8] / q[9]);
            }
    }
#endif
#ifdef TT_AB
    if (TT_AB_SWITCH(TT_GRU_SPLIT_BWD, 1) == 4) // the four-wave member
        hipLaunchKernelGGL(gru_bwd16x4_kernel, dim3((sp.nteams + 7) / 8 * 32, ndir), dim3(256), XB_LDS + 16, st, sp);
    else
#endif
    if (gru16x4_launches(bp.B, bp.H, ndir) == 2) { // one direction after the other, each with its own exchange slots
        for (int d = 0; d < 2; ++d) {
            GruSplitBwdParams one = sp;
            one.g.dir[0] = bp.dir[d];
            one.xch = (char *)xch + (size_t)d * gru16x4_bwd_xch_bytes(bp.B, bp.H, 1);
            hipLaunchKernelGGL(gru_bwd16x4p_kernel, dim3((sp.nteams + 7) / 8 * 32, 1), dim3(512), XP_LDS + 16, st, one);
        }
    } else
        hipLaunchKernelGGL(gru_bwd16x4p_kernel, dim3((sp.nteams + 7) / 8 * 32, ndir), dim3(512), XP_LDS + 16, st, sp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int gru16x4_launch(const GruParams &gp, int ndir, void *xch, int32_t *status, hipStream_t st, bool xch_zeroed)
{
    GruSplitParams sp;
    sp.g = gp;
    sp.xch = (char *)xch;
    sp.status = status;
    sp.nteams = (gp.B + ENC_RB - 1) / ENC_RB;
    sp.spin_max = 1u << 17; // ~0.3 s of backed-off sweeps: a partner that is merely waiting for a CU arrives long before that
    if (!xch_zeroed) // (the first layer's slots are cleared by the call's one zero launch)
        TT_RC_CHECK(tt_zero_async(xch, gru16x4_xch_bytes(gp.B, gp.H, ndir), st));
    static bool attr_done = false;
    if (!attr_done) {
#ifdef TT_AB
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16x4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X4_LDS + 16));
#endif
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16x4p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XF_LDS + 16));
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16x4p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XF_LDS + 16));
        attr_done = true;
    }
#ifdef TT_X4_DBG
    static int calls = 0;
    if (++calls % 8 == 0) {
        unsigned long long h[16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(x4_dbg), sizeof h) == hipSuccess && h[6] && h[14])
            for (int k = 0; k < 2; ++k)
                fprintf(stderr, "x4dbg half %d: clocks per step: matrix + hand-over %.0f, gates %.0f, publish %.0f, sweep %.0f, unpack + stores %.0f, barrier %.0f\n", k,
                        (double)h[8 * k] / h[8 * k + 6], (double)h[8 * k + 1] / h[8 * k + 6], (double)h[8 * k + 2] / h[8 * k + 6],
                        (double)h[8 * k + 3] / h[8 * k + 6], (double)h[8 * k + 4] / h[8 * k + 6], (double)h[8 * k + 5] / h[8 * k + 6]);
    }
#endif
#ifdef TT_AB
    if (TT_AB_SWITCH(TT_GRU_SPLIT, 1) == 4) // the four-wave member
        hipLaunchKernelGGL(gru_seq16x4_kernel, dim3((sp.nteams + 7) / 8 * 32, ndir), dim3(256), X4_LDS + 16, st, sp);
    else
#endif
    {
        if (gp.gi_ids && gp.gi_rows == 0)
            return tt_fail(TT_ERR_BAD_SHAPE, "gru16x4_launch: projected table without rows");
        void (*const kern)(GruSplitParams) = gp.gi_ids ? gru_seq16x4p_kernel<true> : gru_seq16x4p_kernel<false>;
        if (gru16x4_launches(gp.B, gp.H, ndir) == 2) { // one direction after the other, each with its own exchange slots
            for (int d = 0; d < 2; ++d) {
                GruSplitParams one = sp;
                one.g.dir[0] = gp.dir[d];
                one.xch = (char *)xch + (size_t)d * gru16x4_xch_bytes(gp.B, gp.H, 1);
                hipLaunchKernelGGL(kern, dim3((sp.nteams + 7) / 8 * 32, 1), dim3(512), XF_LDS + 16, st, one);
            }
        } else
            hipLaunchKernelGGL(kern, dim3((sp.nteams + 7) / 8 * 32, ndir), dim3(512), XF_LDS + 16, st, sp);
    }
    TT_LAUNCH_CHECK();
    return TT_OK;
}
