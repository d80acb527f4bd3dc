// dsd_loop_rs.hpp - the WHOLE K-step reverse loop as ONE persistent kernel for batches that do NOT fill the chip (gfx950): the row-split loop.
//
// The reference infers ONE utterance per device (configs/tts/fs2.yaml:70 max_eval_sentences: 1; tasks/tts/fs2.py:340-369): 512 frames are 16
// tiles of 32 frames, 16 of 256 CUs for k_loop (dsd_loop.hpp).  dsd_lat.hpp splits the OUTPUT ROWS of a layer's two contractions over G
// workgroups per tile - two kernels per layer, three per head, 43 hipGraph nodes per evaluation whose existence (dispatch, cold operands, drain:
// 4-5 us each) is the cost, not their matrix time.  Here the same row split runs inside ONE launch: the G workgroups of a
// tile stay resident for the whole loop (usr/diff/shallow_diffusion_tts.py:261-270; per step DiffNet.forward usr/diff/net.py:107-130 + p_sample
// :159-166 / p_sample_plms :168-204) and the all-gathers a layer needs - gate rows (net.py:73-74 -> :76), then x' rows (net.py:78 -> next :72) -
// are in-launch exchanges through HBM-side rings instead of kernel boundaries.  Same arithmetic and summation order as the kernels it replaces:
// G = 2 / 4 bit-identical to k_loop / k_layer, G = 8 / 16 bit-identical to k_lat_conv / k_lat_out / k_lat_head_* (tests/test_gpu_rs.py).
//
// THE EXCHANGE - data is the flag, one one-way trip per hop (tools/hop_probe.hip measures it against the flag protocol of k_loop):
//   * a ring has THREE slots; item n (the n-th tile a set of owners publishes into that ring) lives in slot n % 3.  A word that holds the
//     SENTINEL 0xffffffff has not arrived.  (A NaN with every payload bit set: no fp32 operation produces it from other inputs - the hardware's
//     generated NaN is 0x7fc00000 - so it can only appear if the caller's cond / x carry that very NaN; the loop then runs into its spin bound
//     and fails LOUDLY with DSD_ERR_TIMEOUT instead of returning the NaN mel the reference would.)
//   * publish: the owner of a slice stores it write-through (sc1); nothing else - no drain, no flag.  A consumer reads its 16-byte pieces with
//     sc1 loads and re-reads the pieces in which a word is still the sentinel (every word is checked: dword stores are atomic, nothing is
//     assumed about wider ones).  Bounded spins, sticky timeout word, NaN-poisoned result - the contract of k_loop.
//   * reset: before an owner may publish item n it must have RESET its slice of slot (n + 1) % 3 - which holds item n - 2 - to the sentinel and
//     DRAINED those stores (s_waitcnt vmcnt(0)).  Two obligations:
//       (a) nobody reads item n - 2 any more when the reset is issued.  Gate ring: every workgroup of the tile publishes item n - 1 after its
//           own gather of n - 2, and I have gathered n - 1 from all of them.  x ring (readers: the workgroups of my tile AND of both neighbour
//           tiles): every workgroup stages x_l before it publishes gate_l; the residual owners of a tile publish x_{l+1} after gathering gate_l
//           from ALL workgroups of their tile; so once I have staged item n - 1 (my tile's slices and the halo frames of both neighbours,
//           which touch every owner there) every reader is done with item n - 2.  Per-evaluation rings (layer-0 input, skip sum, head
//           tiles): reset at layer 1 of evaluation e for evaluation e + 2 - everything of evaluation e - 1 has been consumed by then.
//       (b) no reader can mistake the OLD contents of the slot (item n - 2) for item n + 1: a reader polls for item n + 1 only after it has
//           consumed my item n, which I stored after the drain of the reset.  Hence the order reset -> drain -> publish n, and three slots:
//           item n - 1 may still be read by a slower neighbour while n is written and the slot of n + 1 is already being cleared.
//   * who can be how far ahead: a workgroup cannot finish phase k before all producers it reads have published phase k, so two workgroups
//     that exchange data are at most one item apart in every ring - the same bound as the parity double buffer of k_loop, plus the slot
//     that is being cleared.
//   * every spin is bounded; all workgroups of a launch must be co-resident: ntiles * G <= CU count, one workgroup per CU (LDS 110 KiB).
// Placement: the G workgroups of a tile sit behind one L2 (lat_map); results do not depend on it.
#pragma once
#include "dsd_lat.hpp"
#include "dsd_loop.hpp"

namespace dsd {

constexpr unsigned kRsSentinel = 0xffffffffu;
constexpr int kRsLDP = kMPad + 4;           // row stride of the frame-major mel tile: 25 x 16 bytes
constexpr int kRsLdsBytes = (kFmY + kFmG + 4 * 32 * 32 + 32 * kRsLDP) * (int)sizeof(float);

struct RsParams {
    const float4* w1p;          // [L][w4][kc96][mb4][lane64]
    const float4* w1q;          // [L][b16][kc96][lane64]                     (G = 16)
    const float4* w2p;          // [L][w4][kc32][mb4][lane64]
    const float* b2raw;         // [L][2C]
    const float4* cp;           // [L][tile][w4][mb4][q4][lane64]
    size_t cp_lstride;
    const float* ds_table;      // [t][L][C]
    int L, T, TS, ntile32, ntiles;
    unsigned char dil[kLoopMaxLayers];
    HeadParams head;
    const HeadParams* evals;    // [n_evals] (device)
    const int* eval_t;          // [n_evals]
    int n_evals;
    const float* spec0;         // [B][M][T] x at loop entry
    // exchange rings, all sentinel-filled at launch: [3 slots][ntiles][32 frames][256] (pb: [..][96])
    float* xb;                  // x_l, l >= 1: item e (L - 1) + l - 1, owners = the waves that finish residual rows
    float* x0b;                 // x_0 of evaluation e (input projection), owners = the input-projection waves
    float* gb;                  // gate tile of (e, l): item e L + l, owners = every workgroup (its gate rows)
    float* sb;                  // skip sum of evaluation e, owners = the waves that finish skip rows
    float* hb;                  // relu(skip_projection) of evaluation e
    float* pb;                  // next x (mel rows, zero padded to 96) of evaluation e
    unsigned* tmo;              // sticky timeout word, zero at launch
    unsigned long long* dbg;    // optional s_memtime stamps [workgroup][wave][16] of layer phase dbg_phase (= evaluation * L + layer)
    int dbg_phase;
};

__device__ __forceinline__ bool rs_is_sentinel(const float4& v) {
    const unsigned a = __float_as_uint(v.x), b = __float_as_uint(v.y), c = __float_as_uint(v.z), d = __float_as_uint(v.w);
    return max(max(a, b), max(c, d)) == kRsSentinel;
}
__device__ __forceinline__ void rs_st16(float* base_uniform, int float_off, const float4& v) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ f = {v.x, v.y, v.z, v.w};
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), r, float_off * 4, 0, 16);       // aux 16 = sc1
}
__device__ __forceinline__ void rs_st8(float* base_uniform, int float_off, float a, float b) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ u = {__float_as_uint(a), __float_as_uint(b)};
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, float_off * 4, 0, 16);
}

// Consume N 16-byte pieces (float offsets off[i] behind the wave-uniform slot base; need[i] false: not mine to read): re-read until no
// word of a needed piece is the sentinel.  Bounded; a timeout is sticky (the loop then finishes on garbage and poisons the result).
template <int N>
__device__ __forceinline__ void rs_gather(const float* base_uniform, const int (&off)[N], const bool (&need)[N], float4 (&v)[N], unsigned* tmo) {
    for (int spins = 0;; ++spins) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (need[i]) v[i] = ld16_sc1(base_uniform, off[i] * 4);
        bool miss = false;
#pragma unroll
        for (int i = 0; i < N; ++i) miss |= need[i] && rs_is_sentinel(v[i]);
        if (!miss) break;
        if ((spins & 255) == 255 && __hip_atomic_load((gu32*)tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins >= kLoopSpinLimit) { __hip_atomic_store((gu32*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// B functor of the dilated conv over the frame-major y tile for a wave whose K range starts at chunk kbase (a multiple of 6): the select form
// of ConvBT (a run-time kbase must not turn the chunk -> pointer map into a branch per chunk: ConvB<LD, true>, dsd_kernels.hpp)
struct ConvBTK {
    const float* yc; int dilrow, kbase;
    __device__ __forceinline__ const float* operator()(int it, int u) const {
        const int kc = kbase + 6 * it + u;
        const int idx = kc - kConvCentre;
        const int oc = kc * 8, oo = (idx >> 1) * 8 + ((idx & 1) ? dilrow : -dilrow);
        return yc + ((kc < kConvCentre) ? oc : oo);
    }
};

// MODE: HEAD_DDPM or HEAD_PLMS; G: workgroups per 32-frame tile
template <int G, int MODE>
__global__ __launch_bounds__(kThreads, 1) void k_loop_rs(const RsParams p) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "row split");
    constexpr int LDK = kFmLDK, LDP = kRsLDP;
    constexpr int NMB = (G == 2) ? 2 : 1;
    constexpr int NCH1 = (G == 16) ? 24 : (G == 8) ? 48 : 96;      // conv chunks per wave
    constexpr int NCH2 = (G == 16) ? 8 : (G == 8) ? 16 : 32;       // out-projection chunks per wave
    constexpr int ASTR1 = (G == 16) ? 64 : 256;
    constexpr int KSH = (G >= 8) ? 4 : 1;                          // K split of the head's contractions over the waves (as k_lat_head_a / _b)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                    // [48][260] conv input y = x + step_proj (+ halo rows), frame-major; head: scaled skip sum [32][260]
    float* gtile = smem + kFmY;             // [32][260] gate tile; head: relu(skip_projection)
    float* red = gtile + kFmG;              // [4][32][32] K partials / filter hand-over
    float* ptile = red + 4 * 1024;          // [32][100] spec / next-x tile of the input projection

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile, g;
    if (!lat_map<G>(p.ntiles, tile, g)) return;
    const int b = tile / p.ntile32, tn = tile - b * p.ntile32, t0 = tn * 32;
    const bool has_left = tn > 0, has_right = tn + 1 < p.ntile32;
    const int T = p.T, M = p.head.M, L = p.L;
    const size_t slotf = (size_t)p.ntiles * (32 * kC), slotp = (size_t)p.ntiles * (32 * kMPad);
    const int tbase = tile * (32 * kC);     // float offset of my tile inside a slot of the 256-wide rings
    const int c4 = tid & 63, fr0 = tid >> 6; // staging: thread -> (frame fr0 + 4 i, channels 4 c4 ..)
    const bool stamp = p.dbg != nullptr;

    // ---- roles of this wave ------------------------------------------------------------------------------------------------------
    // conv (k_lat_conv<G>): packed stream w4, first row block mb0, first chunk; out-projection (k_lat_out<G>) likewise
    int w4, mb0c, kbeg1, mb0o, kbeg2 = 0;
    if (G == 2) { w4 = 2 * g + (wv >> 1); mb0c = wv & 1; kbeg1 = 0; mb0o = wv & 1; }
    else if (G == 4) { w4 = g; mb0c = wv; kbeg1 = 0; mb0o = wv; }
    else if (G == 8) { w4 = g >> 1; mb0c = (g & 1) + 2 * (wv & 1); kbeg1 = 48 * (wv >> 1); mb0o = (g & 1) + 2 * (wv & 1); kbeg2 = 16 * (wv >> 1); }
    else { w4 = g >> 2; mb0c = (g & 3) >> 1; kbeg1 = 24 * wv; mb0o = g & 3; kbeg2 = 8 * wv; }
    // the 32-row blocks this WAVE finishes in the out-projection: residual block rblk and / or skip block sblk of stream w4.  These waves OWN
    // those rows for the whole loop: x (xq) and the running skip sum (skp) of their rows stay in registers, fragment order.
    bool fin_res, fin_skip;
    int rblk = 0, sblk = 0;
    if (G == 2) { fin_res = fin_skip = true; rblk = sblk = wv & 1; }
    else if (G == 4) { fin_res = wv < 2; fin_skip = wv >= 2; rblk = wv & 1; sblk = wv & 1; }
    else if (G == 8) { fin_res = wv == 0; fin_skip = wv == 1; rblk = sblk = g & 1; }
    else { fin_res = wv == 0 && mb0o < 2; fin_skip = wv == 0 && mb0o >= 2; rblk = sblk = mb0o & 1; }
    const int cres = 64 * w4 + 32 * rblk + 4 * h, csk = 64 * w4 + 32 * sblk + 4 * h;       // channel of xq[0].x / skp[0].x (quad q: + 8 q)
    // head (k_lat_head_a / _b / _c for G >= 8, k_head's single chains for G <= 4): row block of the skip projection / input projection
    // (8 blocks) and of the final projection (3 blocks) this wave works on, -1: none
    int hblk = -1, oblk = -1, kh = 0;
    if (G >= 8) { if (g < 8) hblk = g; if (g < 3) oblk = g; kh = wv; }
    else if (G == 4) { if (wv < 2) hblk = 2 * g + wv; if (g < 3 && wv == 0) oblk = g; }
    else { hblk = 4 * g + wv; if (g == 0 && wv < 2) oblk = wv; if (g == 1 && wv == 0) oblk = 2; }
    const bool hfin = hblk >= 0 && (KSH == 1 || wv == 0);            // finishes rows of head_a / head_c (wave 0 of a K-split group)

    float4 xq[4], skp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xq[q] = skp[q] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto ring = [&](float* base, int item) -> float* { return base + (size_t)(item % 3) * slotf; };
    const float4 sent4 = make_float4(__uint_as_float(kRsSentinel), __uint_as_float(kRsSentinel), __uint_as_float(kRsSentinel), __uint_as_float(kRsSentinel));
    // my slice of a 256-wide ring slot as the residual / skip / head-row owner: 4 pieces (quads) of frame j per lane
    auto put_rows = [&](float* slot, int ch0, const float4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs_st16(slot, tbase + j * kC + ch0 + 8 * q, v[q]);
    };
    auto reset_rows = [&](float* slot, int ch0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs_st16(slot, tbase + j * kC + ch0 + 8 * q, sent4);
    };
    // my gate rows in the gate ring (the conv epilogue's store pattern), value or sentinel
    //   G = 16: registers 2 wv, 2 wv + 1 of the half block: channels 16 g + 8 (wv >> 1) + 4 h + 2 (wv & 1) + {0, 1}: one 8-byte store
    //   G = 8 : quad wv of gate block (g & 1): channels 64 w4 + 32 (g & 1) + 8 wv + 4 h + {0..3}: one 16-byte store
    //   G = 4 / 2: the whole gate block on the gate waves (G = 4: waves 0, 1; G = 2: every wave, block mb0c): four 16-byte stores
    const int gch = (G == 16) ? 16 * g + 8 * (wv >> 1) + 4 * h + 2 * (wv & 1) : (G == 8) ? 64 * w4 + 32 * (g & 1) + 8 * wv + 4 * h : 64 * w4 + 32 * mb0c + 4 * h;
    const bool gate_owner = (G == 4) ? (wv < 2) : true;
    auto reset_gate = [&](float* slot) {
        if (G == 16) rs_st8(slot, tbase + j * kC + gch, sent4.x, sent4.x);
        else if (G == 8) rs_st16(slot, tbase + j * kC + gch, sent4);
        else if (gate_owner) reset_rows(slot, gch);
    };
#define RS_STAMP(i) do { if (stamp && ph == p.dbg_phase && lane == 0) p.dbg[((size_t)blockIdx.x * 4 + wv) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    // ---- input projection of evaluation e: next-x tile [32][100] in ptile -> relu(W x + b) rows of block hblk -> x0 ring -----------------
    auto inproj_publish = [&](int e) {
        if (hfin) {
            GemmPipe<1, 1, LDP, 128, 6, TileBT, 1, true> pipe(p.head.winp + (size_t)(hblk >> 1) * p.head.nk_in * 128 + (hblk & 1) * 64, lane, p.head.nk_in,
                                                               TileBT{ptile + j * LDP + 4 * h, p.head.nk_in});
            pipe.start_a();
            f32x16 acc[1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) set4(acc[0][0], q, p.head.binp[(hblk * 2 + h) * 4 + q]);
            pipe.start_b();
            pipe.run(acc, 0, p.head.nk_in);
            float4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = get4(acc[0][0], q);
                o[q] = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            put_rows(ring(p.x0b, e), 32 * hblk + 4 * h, o);
        }
    };

    // evaluation 0: the spec tile comes from global memory
    for (int idx = tid; idx < 32 * LDP; idx += kThreads) ptile[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < M * 32; idx += kThreads) {
        const int m = idx >> 5, f = idx & 31;
        if (t0 + f < T) ptile[f * LDP + m] = p.spec0[((size_t)b * M + m) * T + t0 + f];
    }
    __syncthreads();
    inproj_publish(0);

    int ph = 0;
    for (int e = 0; e < p.n_evals; ++e) {
        const int t_e = p.eval_t[e];
        float4 dsv = *reinterpret_cast<const float4*>(p.ds_table + ((size_t)t_e * L + 0) * kC + 4 * c4);     // step projection of layer 0 at my staging channels
        for (int l = 0; l < L; ++l, ++ph) {
            const bool last = (l == L - 1);
            const int dil = p.dil[l];
            RS_STAMP(0);
            // ===== conv phase =========================================================================================================
            // the weight stream depends on nothing computed here
            const ConvBTK bof1{ytile + (kHalo + j) * LDK + 4 * h, dil * LDK, kbeg1};
            const float4* abase1 = (G == 16) ? p.w1q + ((size_t)l * 16 + g) * (96 * 64) + (size_t)kbeg1 * 64
                                             : p.w1p + ((size_t)l * 4 + w4) * (96 * 256) + (size_t)kbeg1 * 256 + mb0c * 64;
            GemmPipe<NMB, 1, LDK, ASTR1, 6, ConvBTK, 2, true> pipe1(abase1, lane, NCH1, bof1);
            pipe1.start_a();
            // hoisted conditioner projection (+ biases) of the rows this wave finishes (k_lat_conv's indexing)
            const float4* cpl = p.cp + (size_t)l * p.cp_lstride + ((size_t)tile * 4 + w4) * (4 * 4 * 64) + lane;
            float4 cv[4], cf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[q] = cf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (G == 16) {
                cv[0] = cpl[(mb0c * 4 + (wv >> 1) + 2 * (g & 1)) * 64];
                cf[0] = cpl[((mb0c + 2) * 4 + (wv >> 1) + 2 * (g & 1)) * 64];
            } else if (G == 8) {
                cv[0] = cpl[((g & 1) * 4 + wv) * 64];
                cf[0] = cpl[(((g & 1) + 2) * 4 + wv) * 64];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cv[q] = cpl[(mb0c * 4 + q) * 64];
                    if (G == 2) cf[q] = cpl[((mb0c + 2) * 4 + q) * 64];
                }
            }
            // gather x_l: my tile (8 pieces per thread) + dil frames of each neighbour (net.py:69-71: the conv pads y = x + step with zeros)
            const float* xslot = (l == 0) ? ring(p.x0b, e) : ring(p.xb, e * (L - 1) + l - 1);
            {
                int off[12]; bool need[12]; float4 v[12];
                const int hsh = 6 + (31 - __builtin_clz(dil));              // 64 dil pieces per side
#pragma unroll
                for (int i = 0; i < 8; ++i) { off[i] = tbase + (fr0 + 4 * i) * kC + 4 * c4; need[i] = true; }
                int hrow[4], htf[4]; bool hval[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hp = tid + 256 * i;
                    hval[i] = hp < (2 << hsh);
                    const int side = hp >> hsh, fo = (hp & ((1 << hsh) - 1)) >> 6;
                    const bool have = side ? has_right : has_left;
                    off[8 + i] = (side ? tbase + 32 * kC + fo * kC : tbase - dil * kC + fo * kC) + 4 * c4;
                    need[8 + i] = hval[i] && have;
                    hrow[i] = side ? kHalo + 32 + fo : kHalo - dil + fo;
                    htf[i] = side ? t0 + 32 + fo : t0 - dil + fo;
                    v[8 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                rs_gather<12>(xslot, off, need, v, p.tmo);
                RS_STAMP(1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = fr0 + 4 * i;
                    *reinterpret_cast<float4*>(ytile + (kHalo + f) * LDK + 4 * c4) = fm_add_masked(v[i], dsv, t0 + f < T);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (hval[i]) *reinterpret_cast<float4*>(ytile + hrow[i] * LDK + 4 * c4) = fm_add_masked(v[8 + i], dsv, need[8 + i] && htf[i] < T);
            }
            // step projection of the next phase at my staging channels
            {
                const bool more = !last || (e + 1 < p.n_evals);
                const int tn_ = last ? p.eval_t[min(e + 1, p.n_evals - 1)] : t_e, ln_ = last ? 0 : l + 1;
                if (more) dsv = *reinterpret_cast<const float4*>(p.ds_table + ((size_t)tn_ * L + ln_) * kC + 4 * c4);
            }
            __syncthreads();
            RS_STAMP(2);
            // The WHOLE workgroup has seen x_l now (barrier): the residual owners fetch their rows of x_0 (written by the input-projection waves),
            // and everything of evaluation e - 1 and item l - 2 of the x ring has been consumed (header, obligation (a)): clear the slots their
            // successors will use.  The stores drain under the contraction; the publishes wait for them.
            if (l == 0 && fin_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xq[q] = ld16_sc1(xslot, (tbase + j * kC + cres + 8 * q) * 4);
            }
            if (l >= 1 && fin_res) reset_rows(ring(p.xb, e * (L - 1) + l - 1 + 2), cres);
            if (l == 1) {
                if (hfin) reset_rows(ring(p.x0b, e + 2), 32 * hblk + 4 * h);
                if (fin_skip) reset_rows(ring(p.sb, e + 2), csk);
                if (hfin) reset_rows(ring(p.hb, e + 2), 32 * hblk + 4 * h);
                if (oblk >= 0) {
                    // my pieces of the next-x ring: frame j, mel quads of block oblk (G >= 8: quad wv; else all four)
                    float* ps = p.pb + (size_t)((e + 2) % 3) * slotp;
                    if (KSH == 4) rs_st16(ps, tile * (32 * kMPad) + j * kMPad + 32 * oblk + 8 * wv + 4 * h, sent4);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rs_st16(ps, tile * (32 * kMPad) + j * kMPad + 32 * oblk + 8 * q + 4 * h, sent4);
                    }
                }
            }
            f32x16 acc[NMB][1];
#pragma unroll
            for (int m = 0; m < NMB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
            pipe1.start_b();
            pipe1.run(acc, 0, NCH1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // my sentinel stores have landed (obligation (b)); nothing else is in flight but
            RS_STAMP(3);                                                 // the weight prefetch past the end of the stream

            // out-projection weight stream: requested before the gate exchange
            const float4* abase2 = p.w2p + ((size_t)l * 4 + w4) * (32 * 256) + (size_t)kbeg2 * 256 + mb0o * 64;
            GemmPipe<NMB, 1, LDK, 256, 6, TileBT, 2, true> pipe2(abase2, lane, NCH2, TileBT{gtile + j * LDK + 4 * h + 8 * kbeg2, NCH2});
            const bool out_active = (G == 2) || !(last && mb0o < 2);      // the last layer's residual half is dead (net.py:126 reads the skips only)
            if (out_active) pipe2.start_a();

            // gate (net.py:73-74) of my rows -> gate ring, item e L + l
            float* gslot = ring(p.gb, e * L + l);
            if (G == 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
                __syncthreads();
                float gq[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = 2 * wv + q;
                    const float* pg = red + frag_row(r, h) * 32 + j;
                    const float* pf = red + frag_row(r + 8, h) * 32 + j;
                    const float ag = ((pg[0] + pg[1024]) + pg[2048]) + pg[3072];
                    const float af = ((pf[0] + pf[1024]) + pf[2048]) + pf[3072];
                    gq[q] = sigmoid_f(ag + f4at(cv[0], r & 3)) * tanh_f(af + f4at(cf[0], r & 3));
                }
                rs_st8(gslot, tbase + j * kC + gch, gq[0], gq[1]);
            } else if (G == 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
                __syncthreads();
                float gq[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float* pr = red + frag_row(4 * wv + qq, h) * 32 + j;
                    const float ag = pr[0] + pr[2048], af = pr[1024] + pr[3072];
                    gq[qq] = sigmoid_f(ag + f4at(cv[0], qq)) * tanh_f(af + f4at(cf[0], qq));
                }
                rs_st16(gslot, tbase + j * kC + gch, make_float4(gq[0], gq[1], gq[2], gq[3]));
            } else if (G == 4) {
                float* fx = red + (wv & 1) * 1024;
                if (wv >= 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) fx[frag_row(r, h) * 32 + j] = tanh_f(acc[0][0][r] + f4at(cv[r >> 2], r & 3));
                }
                __syncthreads();
                if (wv < 2) {
                    float4 go[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t4[4];
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) {
                            const int r = 4 * q + ee;
                            t4[ee] = sigmoid_f(acc[0][0][r] + f4at(cv[q], ee)) * fx[frag_row(r, h) * 32 + j];
                        }
                        go[q] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                    }
                    put_rows(gslot, gch, go);
                }
            } else {
                float4 go[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t4[4];
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) {
                        const int r = 4 * q + ee;
                        t4[ee] = sigmoid_f(acc[0][0][r] + f4at(cv[q], ee)) * tanh_f(acc[NMB - 1][0][r] + f4at(cf[q], ee));
                    }
                    go[q] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
                put_rows(gslot, gch, go);
            }
            RS_STAMP(4);

            // ===== out phase ==========================================================================================================
            float4 bq[4];
            if (fin_res && !last) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(p.b2raw + (size_t)l * 2 * kC + cres + 8 * q);
            }
            {
                int off[8]; bool need[8]; float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { off[i] = tbase + (fr0 + 4 * i) * kC + 4 * c4; need[i] = true; }
                rs_gather<8>(gslot, off, need, v, p.tmo);
                RS_STAMP(5);
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(gtile + (fr0 + 4 * i) * LDK + 4 * c4) = v[i];
            }
            __syncthreads();
            RS_STAMP(6);
            // every workgroup of the tile has published gate item e L + l (the whole workgroup has seen it: barrier), i.e. is done with item
            // e L + l - 1: clear the slot item e L + l + 2 will use
            reset_gate(ring(p.gb, e * L + l + 2));
            f32x16 acc2[NMB][1];
#pragma unroll
            for (int m = 0; m < NMB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[m][0][r] = 0.f;
            if (out_active) {
                pipe2.start_b();
                pipe2.run(acc2, 0, NCH2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the sentinel stores of this layer (x ring, gate ring) have landed
            RS_STAMP(7);
            // K partials -> the finishing wave (k_lat_out's orders: G = 16 ((p0 + p1) + p2) + p3 on wave 0, G = 8 first half + second half)
            if (G == 16) {
                lat_ksum4(acc2[0][0], red, wv, j, h);
            } else if (G == 8) {
                float* part = red + (wv & 1) * 1024;
                if (wv >= 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[frag_row(r, h) * 32 + j] = acc2[0][0][r];
                }
                __syncthreads();
                if (wv < 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[0][0][r] = acc2[0][0][r] + part[frag_row(r, h) * 32 + j];
                }
            }
            if (fin_res && !last) {
                // x' = (x + (res + b)) / sqrt(2)   (net.py:78; the operation order of k_layer / k_lat_out) -> x ring item e (L - 1) + l, and kept
                constexpr float kInvSqrt2 = 1.0f / 1.41421354f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = get4(acc2[0][0], q), x = xq[q], bv = bq[q];
                    xq[q] = make_float4((x.x + (v.x + bv.x)) * kInvSqrt2, (x.y + (v.y + bv.y)) * kInvSqrt2, (x.z + (v.z + bv.z)) * kInvSqrt2,
                                        (x.w + (v.w + bv.w)) * kInvSqrt2);
                }
                put_rows(ring(p.xb, e * (L - 1) + l), cres, xq);
            }
            if (fin_skip) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = get4(acc2[NMB - 1][0], q), s = skp[q];
                    skp[q] = (l == 0) ? a : make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
                }
            }
            RS_STAMP(8);
        }

        // ===== head (net.py:126-129) + sampler update + the next evaluation's input projection ==========================================
        HeadParams hp = p.evals[e];
        const bool fuse = (e + 1 < p.n_evals);
        float* stile = ytile;               // [32][260] scaled skip sum, frame-major
        float* htile = gtile;               // [32][260] relu(skip projection)
        // skip sum -> ring (raw sums; the bias and the 1 / sqrt(L) are applied at staging, like k_head / k_lat_head_a)
        if (fin_skip) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            put_rows(ring(p.sb, e), csk, skp);
        }
        // --- a: relu(skip_projection(sum(skip) / sqrt(L))) rows of block hblk -> hb ring
        {
            const int kb = (KSH == 4) ? 8 * kh : 0, nch = (KSH == 4) ? 8 : 32;
            GemmPipe<1, 1, LDK, 128, 6, TileBT, 1, true> pipe_s(p.head.wsp + (size_t)(max(hblk, 0) >> 1) * (32 * 128) + (size_t)kb * 128 + (max(hblk, 0) & 1) * 64,
                                                                lane, nch, TileBT{stile + j * LDK + 4 * h + 8 * kb, nch});
            if (hblk >= 0) pipe_s.start_a();
            if (hblk >= 0 || G < 8) {       // (uniform per workgroup: for G >= 8 the workgroups g >= 8 have no rows here)
                int off[8]; bool need[8]; float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { off[i] = tbase + (fr0 + 4 * i) * kC + 4 * c4; need[i] = true; }
                rs_gather<8>(ring(p.sb, e), off, need, v, p.tmo);
                // bias of the channels 4 c4 .. 4 c4 + 3 in the packed order [w4][ms][h][q]
                const int c = 4 * c4;
                const float4 bs = p.head.bskp[(((c >> 6) * 2 + ((c >> 5) & 1)) * 2 + ((c >> 2) & 1)) * 4 + ((c >> 3) & 3)];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<float4*>(stile + (fr0 + 4 * i) * LDK + c) =
                        make_float4(__fdiv_rn(v[i].x + bs.x, p.head.sqrt_L), __fdiv_rn(v[i].y + bs.y, p.head.sqrt_L),
                                    __fdiv_rn(v[i].z + bs.z, p.head.sqrt_L), __fdiv_rn(v[i].w + bs.w, p.head.sqrt_L));
            }
            __syncthreads();
            f32x16 acc[1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bz = (hblk >= 0 && (KSH == 1 || wv == 0)) ? p.head.bsp[(hblk * 2 + h) * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
                set4(acc[0][0], q, bz);
            }
            if (hblk >= 0) { pipe_s.start_b(); pipe_s.run(acc, 0, nch); }
            if (KSH == 4) lat_ksum4(acc[0][0], red, wv, j, h); else __syncthreads();
            if (hfin) {
                float4 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = get4(acc[0][0], q);
                    o[q] = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                put_rows(ring(p.hb, e), 32 * hblk + 4 * h, o);
            }
        }
        // --- b: final projection rows of block oblk + sampler update -> x (global) and the next-x ring
        {
            const int kb = (KSH == 4) ? 8 * kh : 0, nch = (KSH == 4) ? 8 : 32;
            const bool ob_wg = (G >= 8) ? (g < 3) : (G == 4) ? (g < 3) : true;          // workgroups with final-projection rows
            GemmPipe<1, 1, LDK, 192, 6, TileBT, 1, true> pipe_o(p.head.woutp + (size_t)kb * 192 + max(oblk, 0) * 64, lane, nch,
                                                                TileBT{htile + j * LDK + 4 * h + 8 * kb, nch});
            if (oblk >= 0) pipe_o.start_a();
            if (ob_wg) {
                int off[8]; bool need[8]; float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { off[i] = tbase + (fr0 + 4 * i) * kC + 4 * c4; need[i] = true; }
                rs_gather<8>(ring(p.hb, e), off, need, v, p.tmo);
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(htile + (fr0 + 4 * i) * LDK + 4 * c4) = v[i];
            }
            __syncthreads();
            const int t = t0 + j;
            // what the sampler update reads besides eps, at the positions this wave finishes (G >= 8: quad wv of the block; else all four quads)
            HeadPre pre[16];
            if (oblk >= 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (KSH == 4 && (r >> 2) != 0) continue;
                    const int rr = (KSH == 4) ? 4 * wv + r : r;
                    const int m = 32 * oblk + frag_row(rr, h);
                    const bool ok = (m < M) && (t < T);
                    const size_t idx = ((size_t)b * M + (ok ? m : 0)) * T + (ok ? t : 0);
                    head_prefetch<MODE>(hp, idx, pre[r]);
                }
            }
            DSD_SB();
            f32x16 acc[1][1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bz = (oblk >= 0 && (KSH == 1 || wv == 0)) ? p.head.boutp[(oblk * 2 + h) * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
                set4(acc[0][0], q, bz);
            }
            if (oblk >= 0) { pipe_o.start_b(); pipe_o.run(acc, 0, nch); }
            float* pslot = p.pb + (size_t)(e % 3) * slotp;
            if (KSH == 4) {
                // every wave adds the four partials for ITS quad in wave order ((p0 + p1) + p2) + p3 and finishes those rows (k_lat_head_b)
                if (oblk >= 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[wv * 1024 + frag_row(r, h) * 32 + j] = acc[0][0][r];
                }
                __syncthreads();
                if (oblk >= 0) {
                    float xn4[4];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int row = frag_row(4 * wv + qq, h), m = 32 * oblk + row;
                        const float* pr = red + row * 32 + j;
                        const float eps = ((pr[0] + pr[1024]) + pr[2048]) + pr[3072];
                        const bool ok = (m < M) && (t < T);
                        const size_t idx = ((size_t)b * M + m) * T + t;
                        xn4[qq] = 0.f;
                        if (ok) xn4[qq] = head_apply<MODE>(hp, eps, idx, pre[qq]);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (fuse) rs_st16(pslot, tile * (32 * kMPad) + j * kMPad + 32 * oblk + 8 * wv + 4 * h, make_float4(xn4[0], xn4[1], xn4[2], xn4[3]));
                }
            } else {
                __syncthreads();
                if (oblk >= 0) {
                    float4 xo[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float xn4[4];
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) {
                            const int r = 4 * q + ee, m = 32 * oblk + frag_row(r, h);
                            const bool ok = (m < M) && (t < T);
                            const size_t idx = ((size_t)b * M + m) * T + t;
                            xn4[ee] = 0.f;
                            if (ok) xn4[ee] = head_apply<MODE>(hp, acc[0][0][r], idx, pre[r]);
                        }
                        xo[q] = make_float4(xn4[0], xn4[1], xn4[2], xn4[3]);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (fuse) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rs_st16(pslot, tile * (32 * kMPad) + j * kMPad + 32 * oblk + 8 * q + 4 * h, xo[q]);
                    }
                }
            }
        }
        // --- c: the next evaluation's input projection (net.py:116-118) -> x0 ring item e + 1
        if (fuse) {
            if (hblk >= 0 || G < 8) {
                // 32 frames x 24 mel quads = 768 pieces, three per thread
                int off[3]; bool need[3]; float4 v[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { off[i] = tile * (32 * kMPad) + (tid + 256 * i) * 4; need[i] = true; }
                rs_gather<3>(p.pb + (size_t)(e % 3) * slotp, off, need, v, p.tmo);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int pc = tid + 256 * i, f = pc / 24, mq = pc - 24 * f;
                    *reinterpret_cast<float4*>(ptile + f * LDP + 4 * mq) = v[i];
                }
            }
            __syncthreads();
            inproj_publish(e + 1);
        }
    }
#undef RS_STAMP
    // a wait that hit its spin bound leaves garbage: make it LOUD - poison this tile of the result with NaN
    if (g == 0 && __hip_atomic_load((gu32*)p.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        float* xo = const_cast<float*>(p.spec0);
        for (int idx = tid; idx < M * 32; idx += kThreads) {
            const int m = idx >> 5, t = t0 + (idx & 31);
            if (t < T) xo[((size_t)b * M + m) * T + t] = __builtin_nanf("");
        }
    }
}

}  // namespace dsd
