// Developer probe (GPU): what does ONE in-launch all-gather among the G workgroups of a tile cost on MI355X?  The question the row-split
// persistent loop (csrc/dsd_loop_rs.hpp) stands or falls with: a layer needs two such exchanges (gate rows, then x' rows), and the kernel
// boundary they replace costs ~4-5 us per node.
//
// 256 workgroups (one per CU), G per tile on one XCD (the lat_map of dsd_lat.hpp).  Per phase every workgroup publishes its slice of a
// [32 frames][256 channels] fp32 tile (256 / G channels of every frame = 32 KiB / G) and then gathers the whole tile (32 KiB) - optionally
// + 2 x `halo` frames of the two neighbour tiles (x exchange of the dilated conv).  Every gathered word is VERIFIED against the value its
// producer must have written in that phase (a stale or torn word is counted), under optional per-workgroup jitter (uneven load).
//
//   PROTO 0  data-is-the-flag: slots rotate mod 3; a word that still holds the SENTINEL (0xffffffff, written by the producer itself one
//            phase ahead and drained before its data stores) has not arrived; the consumer re-reads its own 16-byte pieces until none does.
//            One one-way trip per hop.  sc1 (write-through) stores, sc1 loads.
//   PROTO 1  the protocol of k_loop (dsd_loop.hpp): sc1 payload stores, every storing wave drains, barrier, ONE relaxed agent-scope flag per
//            (tile, slice); the consumer polls the G flags, barrier, then sc1 payload loads.  Two dependent trips per hop.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/hop_probe.bin tools/hop_probe.hip && tools/hop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kSentinel = 0xffffffffu;
constexpr int kSpinLimit = 1 << 20;

__device__ __forceinline__ u32x4 ld16_sc1(const unsigned* base_uniform, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base_uniform), 0, 0x7ffffff0, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);          // aux 16 = sc1
}
__device__ __forceinline__ void st16_sc1(unsigned* base_uniform, unsigned byte_off, u32x4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ unsigned word_of(unsigned phase, unsigned tile, unsigned frame, unsigned ch) {
    return ((phase * 2654435761u) ^ (tile * 40503u + frame * 257u + ch)) & 0x7fffffffu;         // never the sentinel
}
__device__ __forceinline__ bool has_sentinel(u32x4 v) { return max(max(v.x, v.y), max(v.z, v.w)) == kSentinel; }

struct Params {
    unsigned* buf;          // [3 slots][ntiles][32][256]
    unsigned* flags;        // [ntiles][16] (PROTO 1)
    unsigned* err;          // [0] mismatches, [1] timeouts
    unsigned long long* cyc;    // [nwg][2] wall_clock64 at loop start / end (100 MHz)
    int ntiles, phases, work_ticks, jitter_ticks, halo;
};

template <int G, int PROTO>
__global__ __launch_bounds__(256, 1) void k_hop(const Params p) {
    extern __shared__ unsigned hold_lds[];      // 100 KiB requested at launch: one workgroup per CU, like the kernel this probe stands for
    const int tid = threadIdx.x;
    if (p.phases < 0) hold_lds[tid] = tid;
    const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
    const int tile = (k / G) * 8 + xcd, g = k % G;
    if (tile >= p.ntiles) return;
    constexpr int CH = 256 / G;                 // channels of a slice
    constexpr int NPUB = 32 * CH / 4;           // 16-byte pieces of a slice: frame = piece / (CH / 4), quad = piece % (CH / 4)
    const size_t slot_words = (size_t)p.ntiles * 32 * 256;
    const bool has_l = tile > 0, has_r = tile + 1 < p.ntiles;
    unsigned bad = 0, tmo = 0;
    const unsigned long long t_begin = wall_clock64();
    for (int ph = 0; ph < p.phases; ++ph) {
        // "work" of the phase (the contractions), with per-workgroup jitter so that the exchange is tested under uneven load
        if (p.work_ticks + p.jitter_ticks > 0) {
            const unsigned jit = p.jitter_ticks ? ((unsigned)(lin * 2654435761u + ph * 40503u) >> 8) % (unsigned)p.jitter_ticks : 0u;
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < (unsigned long long)(p.work_ticks + jit)) __builtin_amdgcn_s_sleep(1);
        }
        unsigned* slot = p.buf + (size_t)(ph % 3) * slot_words;
        unsigned* mine = slot + (size_t)tile * 32 * 256;
        // ---- publish my slice ------------------------------------------------------------------------------------------------
        if (PROTO == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my sentinel stores of the previous phase have landed
        for (int pc = tid; pc < NPUB; pc += 256) {
            const int f = pc / (CH / 4), q = pc % (CH / 4), ch = CH * g + 4 * q;
            const u32x4 v = {word_of(ph, tile, f, ch), word_of(ph, tile, f, ch + 1), word_of(ph, tile, f, ch + 2), word_of(ph, tile, f, ch + 3)};
            st16_sc1(mine, (unsigned)(f * 256 + ch) * 4u, v);
        }
        if (PROTO == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store((gu32*)(p.flags + tile * 16 + g), (unsigned)ph + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // wait for the G slices of my tile (and of the neighbours when their halo is read)
            if (tid < 48) {
                const int tt = tile + (tid >> 4) - 1, gg = tid & 15;
                const bool need = gg < G && ((tid >> 4) == 1 || (p.halo > 0 && ((tid >> 4) == 0 ? has_l : has_r)));
                if (need) {
                    const gu32* f = (const gu32*)(p.flags + tt * 16 + gg);
                    for (int spins = 0;; ++spins) {
                        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)ph + 1u) break;
                        if (spins >= kSpinLimit) { tmo = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            __syncthreads();
        }
        // ---- gather the whole tile (+ halo frames of the neighbours) ---------------------------------------------------------------
        u32x4 v[8];
        {
            int spins = 0;
            bool todo = true;
            while (todo) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = ld16_sc1(mine, (unsigned)(tid + 256 * i) * 16u);
                todo = false;
                if (PROTO == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) todo |= has_sentinel(v[i]);
                    if (todo && ++spins >= kSpinLimit) { tmo = 1; break; }
                    if (todo) __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = tid + 256 * i, f = pc >> 6, ch = 4 * (pc & 63);
                bad += (v[i].x != word_of(ph, tile, f, ch)) + (v[i].y != word_of(ph, tile, f, ch + 1)) + (v[i].z != word_of(ph, tile, f, ch + 2)) +
                       (v[i].w != word_of(ph, tile, f, ch + 3));
            }
        }
        if (p.halo > 0) {
            // my left halo = the last `halo` frames of tile - 1, my right halo = the first `halo` frames of tile + 1: 64 pieces per frame
            const int npc = 2 * p.halo * 64;
            for (int pc = tid; pc < npc; pc += 256) {
                const int side = pc / (p.halo * 64), r = pc % (p.halo * 64), fo = r >> 6, ch = 4 * (r & 63);
                const bool have = side ? has_r : has_l;
                if (!have) continue;
                const int tt = side ? tile + 1 : tile - 1, f = side ? fo : 32 - p.halo + fo;
                const unsigned* src = slot + (size_t)tt * 32 * 256;
                u32x4 hv;
                for (int spins = 0;; ++spins) {
                    hv = ld16_sc1(src, (unsigned)(f * 256 + ch) * 4u);
                    if (PROTO == 1 || !has_sentinel(hv)) break;
                    if (spins >= kSpinLimit) { tmo = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                bad += (hv.x != word_of(ph, tt, f, ch)) + (hv.y != word_of(ph, tt, f, ch + 1)) + (hv.z != word_of(ph, tt, f, ch + 2)) +
                       (hv.w != word_of(ph, tt, f, ch + 3));
            }
        }
        if (PROTO == 0) {
            // Reset MY slice of the slot phase ph + 2 will use - it holds phase ph - 1.  Every reader of my slice has published ITS phase ph
            // (I have just gathered it) and it did so after its own gather of ph - 1: nobody reads that slot any more.  The sentinel
            // stores drain under the next phase's work, IN FRONT of my phase ph + 1 data stores (s_waitcnt above): whoever sees my phase
            // ph + 1 data - the precondition for polling phase ph + 2 - cannot find stale ph - 1 words in that slot.
            unsigned* nxt = p.buf + (size_t)((ph + 2) % 3) * slot_words + (size_t)tile * 32 * 256;
            const u32x4 s4 = {kSentinel, kSentinel, kSentinel, kSentinel};
            for (int pc = tid; pc < NPUB; pc += 256) {
                const int f = pc / (CH / 4), q = pc % (CH / 4), ch = CH * g + 4 * q;
                st16_sc1(nxt, (unsigned)(f * 256 + ch) * 4u, s4);
            }
        }
    }
    const unsigned long long t_end = wall_clock64();
    if (tid == 0) { p.cyc[2 * lin] = t_begin; p.cyc[2 * lin + 1] = t_end; }
    if (bad) atomicAdd(p.err, bad);
    if (tmo) atomicAdd(p.err + 1, 1u);
}

template <int G, int PROTO>
static int run(int phases, int work_ticks, int jitter_ticks, int halo) {
    const int ntiles = 256 / G;
    const size_t words = (size_t)3 * ntiles * 32 * 256;
    unsigned *buf, *flags, *err;
    unsigned long long* cyc;
    CK(hipMalloc((void**)&buf, words * 4));
    CK(hipMalloc((void**)&flags, (size_t)ntiles * 16 * 4 + 64));
    CK(hipMalloc((void**)&err, 64));
    CK(hipMalloc((void**)&cyc, 256 * 2 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr[2] = {0, 0};
    std::vector<unsigned long long> hc(512);
    double span = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(buf, 0xff, words * 4));
        CK(hipMemset(flags, 0, (size_t)ntiles * 16 * 4 + 64));
        CK(hipMemset(err, 0, 64));
        Params p{buf, flags, err, cyc, ntiles, phases, work_ticks, jitter_ticks, halo};
        CK(hipEventRecord(e0));
        CK(hipFuncSetAttribute((const void*)k_hop<G, PROTO>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        hipLaunchKernelGGL((k_hop<G, PROTO>), dim3(256), dim3(256), 100 * 1024, 0, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned he[2];
        CK(hipMemcpy(he, err, 8, hipMemcpyDeviceToHost));
        herr[0] += he[0]; herr[1] += he[1];
        CK(hipMemcpy(hc.data(), cyc, 512 * 8, hipMemcpyDeviceToHost));
        if (ms < best) {
            best = ms;
            unsigned long long lo = ~0ull, hi = 0;
            for (int i = 0; i < 256; ++i) { lo = hc[2 * i] < lo ? hc[2 * i] : lo; hi = hc[2 * i + 1] > hi ? hc[2 * i + 1] : hi; }
            span = (double)(hi - lo) * 10.0;      // ns (100 MHz)
        }
    }
    const double per = span / phases / 1000.0 - (work_ticks + jitter_ticks * 0.5) * 0.01;
    printf("{\"G\": %d, \"proto\": \"%s\", \"phases\": %d, \"work_us\": %.2f, \"jitter_us\": %.2f, \"halo_frames\": %d, \"kernel_ms\": %.4f, "
           "\"us_per_phase\": %.3f, \"us_per_hop_net_of_work\": %.3f, \"mismatched_words\": %u, \"timeouts\": %u}\n",
           G, PROTO == 0 ? "sentinel (data is the flag, 3 slots)" : "flag (sc1 payload, drain, flag, poll, sc1 loads)", phases, work_ticks * 0.01,
           jitter_ticks * 0.01, halo, best, span / phases / 1000.0, per, herr[0], herr[1]);
    fflush(stdout);
    (void)hipFree(buf); (void)hipFree(flags); (void)hipFree(err); (void)hipFree(cyc);
    return 0;
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    for (int halo : {0, 8}) {
        for (int work : {0, 300}) {                 // 0 and 3 us of "contraction" between the hops
            for (int jit : {0, 100}) {
                if (work == 0 && jit) continue;
                if (run<16, 0>(phases, work, jit, halo)) return 1;
                if (run<16, 1>(phases, work, jit, halo)) return 1;
            }
        }
        if (run<8, 0>(phases, 300, 100, halo)) return 1;
        if (run<8, 1>(phases, 300, 100, halo)) return 1;
        if (run<4, 0>(phases, 300, 100, halo)) return 1;
        if (run<2, 0>(phases, 300, 100, halo)) return 1;
        if (run<2, 1>(phases, 300, 100, halo)) return 1;
    }
    return 0;
}
