// dsd.hip - host side of libdsdenoise.so: handle, weight repacking, schedule tables, workspace, launch
// sequencing of the K-step reverse loop (eager or as a cached hipGraph).  C ABI in include/dsd.h.
#include "dsd_kernels.hpp"
#include "dsd_loop.hpp"
#include "dsd_lat.hpp"
#include "dsd_split.hpp"
#include "dsd_loop_split.hpp"
#include "dsd_loop_wino.hpp"
#include "dsd_lat_wino.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/dsd.h"

using namespace dsd;

// ------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(DSD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define DSD_TRY(expr)            \
    do {                         \
        int r_ = (expr);         \
        if (r_ != DSD_OK) return r_; \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------------------
struct GraphKey {
    int kind, B, T, k_step, interval, tile;
    bool operator<(const GraphKey& o) const {
        return std::tie(kind, B, T, k_step, interval, tile) < std::tie(o.kind, o.B, o.T, o.k_step, o.interval, o.tile);
    }
};

struct dsd_handle {
    dsd_config cfg{};
    int device = 0;
    int L = 0, M = 0, nk_in = 0;
    std::vector<int> dil;
    bool has_weights = false, has_schedule = false, has_spec = false, prepared = false;
    bool use_graph = true;
    int layer_tile_req = 0;     // 0 auto, 32, 64
    int64_t bytes = 0;       // device bytes owned: packed weights + tables (persistent)
    int64_t bytes_ws = 0;    // ... + workspace of the prepared batch

    // packed weights (device)
    float4 *w1p = nullptr, *w2p = nullptr, *wcp = nullptr, *b1p = nullptr, *bskp = nullptr;
    float4* w1q = nullptr;      // dilated conv once more, 16 gate rows + their 16 filter rows per 32-row block (G = 16 latency kernels, dsd_lat.hpp)
    float *b2raw = nullptr, *bsum = nullptr;   // output_projection biases [L][2C]; sum over layers of their skip halves [C]
    float4 *winp = nullptr, *binp = nullptr, *wsp = nullptr, *bsp = nullptr, *woutp = nullptr, *boutp = nullptr;
    // raw copies for the step table
    float *mlp0_w = nullptr, *mlp0_b = nullptr, *mlp2_w = nullptr, *mlp2_b = nullptr, *dp_w = nullptr, *dp_b = nullptr;
    float* ds_table = nullptr;  // [n_table][L][C]
    int n_table = 0;

    // schedule (host fp32 tables, registration order of the reference)
    std::vector<float> tab[12];
    int n_sched = 0;
    float *spec_min_d = nullptr, *spec_max_d = nullptr;

    // workspace for the prepared batch
    int B = 0, T = 0, TS = 0, ntile32 = 0, ntiles = 0;
    int64_t cap_frames = 0;     // capacity (B*ntile32 tiles) the buffers were sized for
    int cap_B = 0;
    int64_t cap_spec = 0;
    float *xa_base = nullptr, *xb_base = nullptr, *xa = nullptr, *xb = nullptr, *condT = nullptr;
    float4 *cp = nullptr, *skip = nullptr;
    float *xs = nullptr, *xtmp = nullptr, *ering[4] = {nullptr, nullptr, nullptr, nullptr};
    int* t_dev = nullptr;
    const float** noise_cell = nullptr;
    unsigned long long* seed_cell = nullptr;
    unsigned long long noise_seed = 0;   // Philox seed used when a sampler is called without explicit noise (dsd_set_noise_seed)

    hipStream_t cap_stream = nullptr;
    std::map<GraphKey, hipGraphExec_t> graphs;

    // pinned host staging ring for the small per-call HOST arrays (dsd_denoise's t[B], dsd_p_sample_ex's coefficients): the caller's
    // array is copied into a slot on the host, the H2D copy from pinned memory is truly asynchronous - no stream synchronisation
    struct PinRing { char* base = nullptr; size_t slot = 0; int next = 0; hipEvent_t ev[8] = {}; bool used[8] = {}; } pin;
    float* coef_dev = nullptr;  // [cap_B][5] per-utterance p_sample coefficients (dsd_p_sample_ex)
    float* eps_tmp = nullptr;   // [cap_spec] eps of dsd_p_sample_ex

    // how the K-step loops run: 0 per-layer kernels (k_layer); 1 the persistent loop (dsd_loop.hpp) whenever the batch geometry allows
    // it; 2 (default) automatic - row-split latency kernels (dsd_lat.hpp) for batches that fill less than half the chip, else the
    // persistent loop unless its whole-utterance chunking wastes more of the chip than the per-layer kernels would; 3 latency kernels
    int loop_mode = 2;
    int lat_req = -1;           // row split of the latency kernels: -1 by batch size, 0 never, 2 / 4 / 8 forced (dsd_set_lat_split)
    float* gbuf = nullptr;      // [ntiles][C][32] gate tiles between k_lat_conv and k_lat_out
    int n_cu = 0;               // workgroups that are certainly co-resident at 1 per CU
    struct LoopPlan { HeadParams* evals = nullptr; int* eval_t = nullptr; int n_evals = 0; };
    std::map<GraphKey, LoopPlan> plans;
    unsigned* loop_flags = nullptr;   // [ntiles] + timeout word behind it
    // the timeout word of every finished persistent loop is latched (k_latch_tmo, enqueued behind the loop) into a word of PINNED host
    // memory: the next call into the handle - or dsd_check - reads it without synchronising anything and fails loudly
    unsigned* sticky_host = nullptr;
    unsigned* sticky_dev = nullptr;
    bool persist_off = false;         // set when a timeout was reported: the handle is PARKED on the hipGraph path ...
    int parked_calls = 0;             // ... for kParkedCalls sampling calls (or until dsd_set_loop_mode), then the persistent path is re-armed
    float* loop_halo = nullptr;       // [2][ntiles][2][256][8]
    int loop_cap_tiles = 0;
    int loop_tmo_at = -1;             // index of the timeout word of the LAST persistent run inside loop_flags (its ntiles), -1: none yet
    unsigned long long* loop_dbg = nullptr;   // debug: stamps of one phase (dsd_debug_loop_timeline)
    int loop_dbg_phase = 0;
    // the dilated convolution of the PERSISTENT loop (dsd_loop_wino.hpp): 1 = Winograd F(2,3) along the frame axis (default), 0 = the direct
    // K = 768 contraction (k_loop: bit-identical to the per-layer kernels).  Every other path evaluates the direct form.
    int conv_mode = 1;
    float4* w1w = nullptr;            // transformed conv weights U0..U3 of all layers in consumption order [L][128 steps][w4][r4][lane64]
    int wino_touch = 32;              // steps (16 KiB each) the L2 touch of that stream runs in front, 0 = off (16 in round 5; 32: 101.3 ms against 102.0 over
                                      // four A/B pairs on one box, profiles/r6_10_wino_ab_touch.jsonl - inside the noise, never behind)
    bool cp_wino = false;             // layout of the prepared batch's cp: the Winograd loop's accumulator order, or the 32x32 fragment order

    // EXPERIMENT (dsd_split.hpp): residual layers on the bf16 matrix pipe with fp32-class accuracy; per-layer kernel path only
    bool split_mode = false;
    uint4 *w1s = nullptr, *w2s = nullptr;     // bf16 weight planes in 32x32x16 fragment order, [L][4][48|16][4][3][64]
    uint4* wlc = nullptr;                     // the planes once more for the persistent split loop (dsd_loop_split.hpp): a layer's 48 conv chunks (centre taps
                                              // first) + 16 out-projection chunks in consumption order, [L][64][wave 4][12 KiB]
    uint4* wl2 = nullptr;                     // the pair format of the split loop: two fp16 planes, consumption order [L][64][wave 4][8 KiB]
    int split_touch = 8;                      // plane stream of the split loop: chunks the L2 touch runs in front, 0 = off (DSD_SPLIT_TOUCH; dsd_loop_split.hpp)
    // format / weight stream of the split loop (DSD_SPLIT_W; dsd_loop_split.hpp): 2 = the PAIR format, two scaled fp16 planes and three products
    // per product (default); 0 = three bf16 planes, six products (the cross-check stream)
    int split_w = 2;
};

// After a reported timeout the handle runs this many sampling loops on the hipGraph path before it tries the persistent loop again: a
// long-lived server that was starved ONCE (a foreign kernel held CUs) must not run the slower path for the rest of its life, and one that is
// starved all the time pays one failed loop in kParkedCalls + 1.
static const int kParkedCalls = 16;
static const int kSlack = 64;   // floats of slack in front of / behind the x buffers (masked halo loads)

template <typename T>
static int dev_alloc(dsd_handle* h, T** p, size_t count, bool workspace = false) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) return fail(DSD_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    *p = (T*)q;
    (workspace ? h->bytes_ws : h->bytes) += (int64_t)(count * sizeof(T));
    return DSD_OK;
}
template <typename T>
static void dev_free(T*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
}

static const int kPinSlots = 8;
// A slot of at least `bytes` of pinned host memory; blocks only if the copy that last read this slot (kPinSlots calls ago) is still
// in flight.  pin_release records the event that marks the slot's H2D copy on `s`.
static int pin_acquire(dsd_handle* h, size_t bytes, char** out, int* slot) {
    auto& r = h->pin;
    if (bytes > r.slot) {
        for (int i = 0; i < kPinSlots; ++i)
            if (r.used[i]) { HIP_TRY(hipEventSynchronize(r.ev[i])); r.used[i] = false; }
        // allocate first, commit base / slot only on success: a failed growth leaves the old (smaller) ring usable
        const size_t slot = (bytes + 4095) / 4096 * 4096;
        char* base = nullptr;
        HIP_TRY(hipHostMalloc((void**)&base, slot * kPinSlots, hipHostMallocDefault));
        if (r.base) (void)hipHostFree(r.base);
        r.base = base;
        r.slot = slot;
        for (int i = 0; i < kPinSlots; ++i)
            if (!r.ev[i]) HIP_TRY(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
    if (!r.base) return fail(DSD_ERR_NOMEM, "pinned staging ring is not allocated");
    const int i = r.next;
    r.next = (r.next + 1) % kPinSlots;
    if (r.used[i]) { HIP_TRY(hipEventSynchronize(r.ev[i])); r.used[i] = false; }
    *out = r.base + (size_t)i * r.slot;
    *slot = i;
    return DSD_OK;
}
static int pin_release(dsd_handle* h, int slot, hipStream_t s) {
    HIP_TRY(hipEventRecord(h->pin.ev[slot], s));
    h->pin.used[slot] = true;
    return DSD_OK;
}

static void drop_graphs(dsd_handle* h) {
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    for (auto& kv : h->plans) { if (kv.second.evals) (void)hipFree(kv.second.evals); if (kv.second.eval_t) (void)hipFree(kv.second.eval_t); }
    h->plans.clear();
}

static void free_workspace(dsd_handle* h) {
    drop_graphs(h);
    dev_free(h->xa_base); dev_free(h->xb_base); dev_free(h->condT); dev_free(h->cp); dev_free(h->skip);
    dev_free(h->xs); dev_free(h->xtmp); dev_free(h->gbuf);
    for (auto& e : h->ering) dev_free(e);
    dev_free(h->t_dev); dev_free(h->coef_dev); dev_free(h->eps_tmp);
    dev_free(h->loop_flags); dev_free(h->loop_halo);
    h->loop_cap_tiles = 0;
    h->loop_tmo_at = -1;
    h->xa = h->xb = nullptr;
    h->cap_frames = 0; h->cap_B = 0; h->cap_spec = 0;
    h->bytes_ws = 0;
    h->prepared = false;
}

#ifndef DSD_BUILD_ID_STR
#define DSD_BUILD_ID_STR "unknown"
#endif
// tag + id in one array: build.py finds the id in the file's bytes without loading the library
#if !defined(__HIP_DEVICE_COMPILE__)       // host side only: the device code object does not change with the id
extern "C" { __attribute__((used, visibility("default"))) const char dsd_build_id_blob[] = "DSD_BUILD_ID=" DSD_BUILD_ID_STR; }
extern "C" const char* dsd_build_id(void) { return dsd_build_id_blob + 13; }
#endif
extern "C" int dsd_abi_version(void) { return DSD_ABI_VERSION; }
extern "C" const char* dsd_last_error(void) { return g_err.c_str(); }

extern "C" int dsd_create(const dsd_config* cfg, int device, dsd_handle** out) {
    if (!cfg || !out) return fail(DSD_ERR_INVALID, "dsd_create: null argument");
    if (cfg->residual_channels != kC || cfg->encoder_hidden != kC)
        return fail(DSD_ERR_INVALID, "dsd_create: this build supports residual_channels == hidden_size == %d (got %d, %d)",
                    kC, cfg->residual_channels, cfg->encoder_hidden);
    if (cfg->mel_bins < 1 || cfg->mel_bins > kMPad) return fail(DSD_ERR_INVALID, "dsd_create: mel_bins must be in 1..%d", kMPad);
    if (cfg->residual_layers < 1 || cfg->residual_layers > 64) return fail(DSD_ERR_INVALID, "dsd_create: residual_layers must be in 1..64");
    if (cfg->dilation_cycle_length < 1) return fail(DSD_ERR_INVALID, "dsd_create: dilation_cycle_length must be >= 1");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DSD_ERR_INVALID, "dsd_create: device %d out of range (%d visible)", device, ndev);
    dsd_handle* h = new dsd_handle();
    h->cfg = *cfg;
    h->device = device;
    h->L = cfg->residual_layers;
    h->M = cfg->mel_bins;
    h->nk_in = (cfg->mel_bins + 7) / 8;
    if (const char* ev = std::getenv("DSD_LOOP")) h->loop_mode = std::atoi(ev);     // the same choice as dsd_set_loop_mode, for an unmodified host
    if (const char* ev = std::getenv("DSD_SPLIT")) h->split_mode = (std::atoi(ev) != 0);     // EXPERIMENT: split-precision layer kernel
    if (const char* ev = std::getenv("DSD_SPLIT_TOUCH")) { const int v = std::atoi(ev); if (v >= 0 && v <= 20) h->split_touch = v; }
    if (const char* ev = std::getenv("DSD_SPLIT_W")) { const int v = std::atoi(ev); if (v == 0 || v == 2) h->split_w = v; }
    if (const char* ev = std::getenv("DSD_CONV")) {                                           // the same choice as dsd_set_conv_mode
        if (!std::strcmp(ev, "direct") || !std::strcmp(ev, "0")) h->conv_mode = 0;
        else if (!std::strcmp(ev, "winograd") || !std::strcmp(ev, "1")) h->conv_mode = 1;
    }
    for (int l = 0; l < h->L; ++l) {
        const int e = l % cfg->dilation_cycle_length;
        if (e > 3) { delete h; return fail(DSD_ERR_INVALID, "dsd_create: dilation 2^%d exceeds the supported maximum %d", e, kHalo); }
        h->dil.push_back(1 << e);
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        // kernels that need more than 64 KiB of dynamic LDS must opt in
        (void)hipFuncSetAttribute((const void*)k_layer<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<1>());
        (void)hipFuncSetAttribute((const void*)k_layer<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<1>());
        (void)hipFuncSetAttribute((const void*)k_layer<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<2>());
        (void)hipFuncSetAttribute((const void*)k_layer<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, layer_lds_bytes<2>());
        (void)hipFuncSetAttribute((const void*)k_loop<HEAD_DDPM>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_loop<HEAD_PLMS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_lat_conv_w<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLatConvWLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_lat_conv_w<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLatConvWLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_lat_conv_w<8>, hipFuncAttributeMaxDynamicSharedMemorySize, kLatConvWLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_loop_wino<HEAD_DDPM, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopWinoLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_loop_wino<HEAD_PLMS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopWinoLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_head<HEAD_EPS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_head<HEAD_DDPM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_head<HEAD_DDPM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_head<HEAD_PLMS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_head<HEAD_PLMS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kHeadLdsBytes);
    }
    if (e != hipSuccess) { delete h; return fail(DSD_ERR_HIP, "dsd_create: %s", hipGetErrorString(e)); }
    *out = h;
    return DSD_OK;
}

extern "C" void dsd_destroy(dsd_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_workspace(h);
    dev_free(h->w1s); dev_free(h->w2s); dev_free(h->wlc); dev_free(h->wl2); dev_free(h->w1w);
    dev_free(h->w1q); dev_free(h->w1p); dev_free(h->w2p); dev_free(h->wcp); dev_free(h->b1p); dev_free(h->bskp); dev_free(h->b2raw); dev_free(h->bsum);
    dev_free(h->winp); dev_free(h->binp); dev_free(h->wsp); dev_free(h->bsp); dev_free(h->woutp); dev_free(h->boutp);
    dev_free(h->mlp0_w); dev_free(h->mlp0_b); dev_free(h->mlp2_w); dev_free(h->mlp2_b); dev_free(h->dp_w); dev_free(h->dp_b);
    dev_free(h->ds_table); dev_free(h->spec_min_d); dev_free(h->spec_max_d);
    dev_free(h->noise_cell); dev_free(h->seed_cell);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->pin.base) (void)hipHostFree(h->pin.base);
    if (h->sticky_host) (void)hipHostFree(h->sticky_host);
    for (auto& e : h->pin.ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int dsd_set_use_graph(dsd_handle* h, int32_t enable) {
    if (!h) return fail(DSD_ERR_INVALID, "null handle");
    h->use_graph = enable != 0;
    return DSD_OK;
}

extern "C" int dsd_set_layer_tile(dsd_handle* h, int32_t frames) {
    if (!h) return fail(DSD_ERR_INVALID, "null handle");
    if (frames != 0 && frames != 32 && frames != 64) return fail(DSD_ERR_INVALID, "dsd_set_layer_tile: frames must be 0, 32 or 64");
    h->layer_tile_req = frames;
    return DSD_OK;
}

extern "C" int64_t dsd_device_bytes(dsd_handle* h) { return h ? h->bytes + h->bytes_ws : 0; }

// Row split G of the latency kernels for the prepared batch, 0 = not on that path.  Automatic mode: the largest G in {16, 8, 4, 2} that
// still gives every workgroup a CU of its own - i.e. batches that leave at least half of the chip idle - and G = 8 for the band above it
// (between half and 5/8 of the CU count in tiles: 129-160 on 256 CUs, e.g. ONE phrase of 4200-5000 frames or 5 x 1024): the persistent loop
// leaves 37-50 % of the CUs without a tile there, 8 x ntiles workgroups in at most five grid waves measured 115-121 ms against its 127 ms
// per K = 100 call (profiles/r47_midsize_paths.jsonl); one more grid wave (163 tiles) and the loop wins again.
static int lat_g(const dsd_handle* h) {
    if (h->split_mode || h->layer_tile_req || h->lat_req == 0 || h->loop_mode < 2) return 0;
    int g = (16 * h->ntiles <= h->n_cu) ? 16 : (8 * h->ntiles <= h->n_cu) ? 8 : (4 * h->ntiles <= h->n_cu) ? 4 : (2 * h->ntiles <= h->n_cu) ? 2 : 0;
    // (the band exists for the DIRECT-convolution loop only: the Winograd loop takes 104 ms per launch and wins it back - profiles/r5_03_shape_sweep.jsonl)
    if (g == 0 && h->loop_mode == 2 && !(h->conv_mode == 1 && h->w1w) && h->ntiles < h->n_cu && 8 * h->ntiles <= 5 * h->n_cu) g = 8;
    if (h->loop_mode == 3 && g == 0) g = 2;
    if (g && (h->lat_req == 2 || h->lat_req == 4 || h->lat_req == 8 || h->lat_req == 16)) g = h->lat_req;
    return g;
}

static int layer_nb(const dsd_handle* h) {
    if (h->split_mode) return 1;               // the split-precision layer kernel exists for 32-frame tiles only
    if (lat_g(h)) return 1;
    if (h->layer_tile_req) return h->layer_tile_req / 32;
    // 32-frame workgroups until there are enough of them to keep two resident per CU on all 256 CUs; beyond
    // that 64-frame workgroups halve the weight traffic out of L2 per frame.
    return (h->ntiles > 1024) ? 2 : 1;
}
extern "C" int dsd_get_layer_tile(dsd_handle* h) { return h ? 32 * layer_nb(h) : 0; }

// Function attributes (dynamic LDS above 64 KiB) belong to a DEVICE's code object: true the first time call site `site` is reached on the
// current device - a process that drives several GPUs (the reference's DP threads, utils/pl_utils.py:146-154) sets them on each.
// site ids in use (one per call site, never shared): 1, 2 dsd.hip; 10, 11 fs2_abi.hpp; 30 train_abi.hpp; 40 voc_abi.hpp (ParallelWaveGAN)
static bool first_on_device(int site) {
    static std::mutex mu;
    static std::set<std::pair<int, int>> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    return seen.emplace(dev, site).second;
}

// ------------------------------------------------------------------------------------------------------------
// loud persistent-loop failures
// ------------------------------------------------------------------------------------------------------------
// One thread, enqueued behind every persistent loop: copies a raised timeout word into the handle's pinned host word (system scope).
// Bit 1 of the word (kLoopRangeBit, dsd_loop_split.hpp: an activation left fp16's range in the pair format) goes to the second pinned word.
__global__ void k_latch_tmo(const unsigned* tmo, unsigned* sticky) {
    const unsigned v = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // both bits independently: a launch can meet a range fault in one tile and a genuine spin-bound timeout in another
    if (v & 2u) __hip_atomic_fetch_add(sticky + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v & ~2u) __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Debug / test hook kernel: holds a CU (all of its LDS) for `ticks` of the 100 MHz wall clock, or until the caller sets the release word
// ctl[1] (ctl[0] counts the holders that are resident).
__global__ void k_hold_cu(unsigned long long ticks, unsigned* ctl) {
    extern __shared__ unsigned hold_lds[];
    hold_lds[threadIdx.x] = threadIdx.x;
    if (ctl && threadIdx.x == 0) atomicAdd(ctl, 1u);                    // the caller can wait until the holders are resident
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        if (ctl && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        __builtin_amdgcn_s_sleep(64);
    }
    if (ctl && hold_lds[threadIdx.x] == 0xffffffffu) *ctl = 0u;          // keeps the LDS allocation alive
}

static int sticky_alloc(dsd_handle* h) {
    if (h->sticky_host) return DSD_OK;
    HIP_TRY(hipHostMalloc((void**)&h->sticky_host, 64, hipHostMallocMapped));
    std::memset(h->sticky_host, 0, 64);
    HIP_TRY(hipHostGetDevicePointer((void**)&h->sticky_dev, h->sticky_host, 0));
    return DSD_OK;
}

// Non-synchronising: has a persistent loop that FINISHED since the last report hit its spin bound?  Consumes the report, parks the
// handle on the hipGraph path (a retry of the same call then runs the per-layer kernels) and fails with DSD_ERR_TIMEOUT.
static int check_sticky(dsd_handle* h, const char* who) {
    if (!h->sticky_host) return DSD_OK;
    // a genuine timeout (bit 0 of a launch's word) is reported first and parks the handle; a range report of the same launches stays latched
    // and is delivered by the next call.  (A range fault makes the other workgroups leave their waits EARLY - that raises no timeout bit.)
    if (__atomic_load_n(h->sticky_host, __ATOMIC_ACQUIRE) == 0u)
    if (const unsigned r = __atomic_load_n(h->sticky_host + 1, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(h->sticky_host + 1, 0u, __ATOMIC_RELEASE);
        return fail(DSD_ERR_RANGE,
                    "%s: %u split-precision loop launch(es) of an EARLIER call on this handle met an activation outside fp16's range (|x| > 65504) - the "
                    "pair format (two fp16 planes per operand, DSD_SPLIT_W=2) cannot hold it; the mel / x tiles those calls returned are NaN.  Turn "
                    "the mode off (dsd_set_split_mode) or use the bf16 planes (DSD_SPLIT_W=0), whose range is fp32's", who, r);
    }
    const unsigned v = __atomic_load_n(h->sticky_host, __ATOMIC_ACQUIRE);
    if (v == 0u) return DSD_OK;
    __atomic_store_n(h->sticky_host, 0u, __ATOMIC_RELEASE);
    h->persist_off = true;
    h->parked_calls = 0;
    return fail(DSD_ERR_TIMEOUT,
                "%s: %u persistent K-step loop launch(es) of an EARLIER call on this handle hit the inter-workgroup spin bound (a foreign kernel held "
                "compute units the loop needs: another process on this GPU, a collective or another model on a side stream) - the mel / x tiles "
                "those calls returned are NaN.  The handle is parked on the hipGraph path (per-layer kernels) for the next %d sampling calls "
                "(dsd_loop_parked; dsd_set_loop_mode re-arms at once): repeat the call", who, v, kParkedCalls);
}

// ------------------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------------------
static int pack_a(hipStream_t s, const float* src, float4* dst, int nw, int ntap, int nkc, int nmb, int split, int hi_base,
                  int rows_valid, int cols_valid, int row_stride, int col_stride) {
    PackParams p{src, (float*)dst, nw, nkc, nmb, ntap, split, hi_base, rows_valid, cols_valid, row_stride, col_stride, /*centre_first*/ 1};
    const size_t n = (size_t)nw * ntap * nkc * nmb * 256;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pack_a, dim3(blocks), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int pack_bias(hipStream_t s, const float* a, const float* b, float4* dst, int nw, int nmb, int split, int hi_base, int rows_valid) {
    PackBiasParams p{a, b, (float*)dst, nw, nmb, split, hi_base, rows_valid};
    hipLaunchKernelGGL(k_pack_bias, dim3((nw * nmb * 32 + 255) / 256), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int build_step_table(dsd_handle* h, int n, hipStream_t s) {
    // ds_table[t][l][c] = diffusion_projection_l( mlp( SinusoidalPosEmb(t) ) )   (net.py:119-120, :67)
    if (n <= h->n_table) return DSD_OK;
    n = (n + 63) / 64 * 64;
    HIP_TRY(hipStreamSynchronize(s));
    drop_graphs(h);
    if (h->ds_table) { h->bytes -= (int64_t)h->n_table * h->L * kC * 4; dev_free(h->ds_table); h->n_table = 0; }
    DSD_TRY(dev_alloc(h, &h->ds_table, (size_t)n * h->L * kC));
    float *E = nullptr, *H1 = nullptr, *H2 = nullptr;
    HIP_TRY(hipMalloc((void**)&E, (size_t)kC * n * 4));
    HIP_TRY(hipMalloc((void**)&H1, (size_t)4 * kC * n * 4));
    HIP_TRY(hipMalloc((void**)&H2, (size_t)kC * n * 4));
    const dim3 blk(64);
    const int gx = (n + 63) / 64;
    hipLaunchKernelGGL(k_step_embed, dim3(gx, kC), blk, 0, s, E, kC, n);
    hipLaunchKernelGGL(k_small_gemm, dim3(gx, 4 * kC), blk, 0, s, h->mlp0_w, h->mlp0_b, E, H1, 4 * kC, kC, n, 1, (size_t)n, (size_t)1);
    hipLaunchKernelGGL(k_small_gemm, dim3(gx, kC), blk, 0, s, h->mlp2_w, h->mlp2_b, H1, H2, kC, 4 * kC, n, 0, (size_t)n, (size_t)1);
    // all layers' diffusion_projection stacked: rows m = l*C + c -> table[t][m]
    hipLaunchKernelGGL(k_small_gemm, dim3(gx, h->L * kC), blk, 0, s, h->dp_w, h->dp_b, H2, h->ds_table, h->L * kC, kC, n, 0,
                       (size_t)1, (size_t)h->L * kC);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(E); (void)hipFree(H1); (void)hipFree(H2);
    if (e != hipSuccess) return fail(DSD_ERR_HIP, "step table: %s", hipGetErrorString(e));
    h->n_table = n;
    return DSD_OK;
}

static void split_kernel_attrs();
static int pack_split_planes(dsd_handle* h, hipStream_t s);

extern "C" int dsd_load_weights(dsd_handle* h, const dsd_weights* w, void* stream) {
    if (!h || !w) return fail(DSD_ERR_INVALID, "dsd_load_weights: null argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int L = h->L, M = h->M;
    if (!h->w1p) {
        DSD_TRY(dev_alloc(h, &h->w1p, (size_t)L * 4 * 96 * 256 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->w1q, (size_t)L * 16 * 96 * 64 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->w2p, (size_t)L * 4 * 32 * 256 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->w1w, (size_t)L * kWnSteps * (kWnStepBytes / 16) + kWeightSlack));
        HIP_TRY(hipMemsetAsync(h->w1w + (size_t)L * kWnSteps * (kWnStepBytes / 16), 0, (size_t)kWeightSlack * 16, s));
        DSD_TRY(dev_alloc(h, &h->wcp, (size_t)L * 4 * 32 * 256 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->b1p, (size_t)L * 4 * 4 * 8));
        DSD_TRY(dev_alloc(h, &h->bskp, (size_t)4 * 2 * 8));
        DSD_TRY(dev_alloc(h, &h->b2raw, (size_t)L * 2 * kC));
        DSD_TRY(dev_alloc(h, &h->bsum, (size_t)kC));
        DSD_TRY(dev_alloc(h, &h->winp, (size_t)4 * h->nk_in * 128 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->binp, (size_t)4 * 2 * 8));
        DSD_TRY(dev_alloc(h, &h->wsp, (size_t)4 * 32 * 128 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->bsp, (size_t)4 * 2 * 8));
        DSD_TRY(dev_alloc(h, &h->woutp, (size_t)32 * 3 * 64 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->boutp, (size_t)3 * 8));
        DSD_TRY(dev_alloc(h, &h->mlp0_w, (size_t)4 * kC * kC));
        DSD_TRY(dev_alloc(h, &h->mlp0_b, (size_t)4 * kC));
        DSD_TRY(dev_alloc(h, &h->mlp2_w, (size_t)4 * kC * kC));
        DSD_TRY(dev_alloc(h, &h->mlp2_b, (size_t)kC));
        DSD_TRY(dev_alloc(h, &h->dp_w, (size_t)L * kC * kC));
        DSD_TRY(dev_alloc(h, &h->dp_b, (size_t)L * kC));
        DSD_TRY(dev_alloc(h, &h->noise_cell, 1));
        DSD_TRY(dev_alloc(h, &h->seed_cell, 1));
    }
    for (int l = 0; l < L; ++l) {
        if (!w->dilated_conv_w[l] || !w->dilated_conv_b[l] || !w->diffusion_projection_w[l] || !w->diffusion_projection_b[l] ||
            !w->conditioner_projection_w[l] || !w->conditioner_projection_b[l] || !w->output_projection_w[l] || !w->output_projection_b[l])
            return fail(DSD_ERR_INVALID, "dsd_load_weights: null tensor in layer %d", l);
        // gate rows [0,C) / filter rows [C,2C) split across waves so each wave owns matching pairs
        DSD_TRY(pack_a(s, w->dilated_conv_w[l], h->w1p + (size_t)l * 4 * 96 * 256, 4, 3, 32, 4, 1, kC, 2 * kC, kC, 3 * kC, 3));
        DSD_TRY(pack_a(s, w->dilated_conv_w[l], h->w1q + (size_t)l * 16 * 96 * 64, 16, 3, 32, 1, 2, kC, 2 * kC, kC, 3 * kC, 3));
        hipLaunchKernelGGL(k_pack_wino, dim3(1024), dim3(256), 0, s, w->dilated_conv_w[l], reinterpret_cast<float*>(h->w1w + (size_t)l * kWnSteps * (kWnStepBytes / 16)));
        DSD_TRY(pack_a(s, w->conditioner_projection_w[l], h->wcp + (size_t)l * 4 * 32 * 256, 4, 1, 32, 4, 1, kC, 2 * kC, kC, kC, 1));
        DSD_TRY(pack_a(s, w->output_projection_w[l], h->w2p + (size_t)l * 4 * 32 * 256, 4, 1, 32, 4, 1, kC, 2 * kC, kC, kC, 1));
        DSD_TRY(pack_bias(s, w->dilated_conv_b[l], w->conditioner_projection_b[l], h->b1p + (size_t)l * 128, 4, 4, 1, kC, 2 * kC));
        HIP_TRY(hipMemcpyAsync(h->b2raw + (size_t)l * 2 * kC, w->output_projection_b[l], (size_t)2 * kC * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->dp_w + (size_t)l * kC * kC, w->diffusion_projection_w[l], (size_t)kC * kC * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->dp_b + (size_t)l * kC, w->diffusion_projection_b[l], (size_t)kC * 4, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_sum_skip_bias, dim3(1), dim3(kC), 0, s, h->b2raw, h->bsum, L);
    HIP_TRY(hipGetLastError());
    DSD_TRY(pack_bias(s, h->bsum, nullptr, h->bskp, 4, 2, 0, 0, kC));
    DSD_TRY(pack_a(s, w->input_projection_w, h->winp, 4, 1, h->nk_in, 2, 0, 0, kC, M, M, 1));
    DSD_TRY(pack_bias(s, w->input_projection_b, nullptr, h->binp, 4, 2, 0, 0, kC));
    DSD_TRY(pack_a(s, w->skip_projection_w, h->wsp, 4, 1, 32, 2, 0, 0, kC, kC, kC, 1));
    DSD_TRY(pack_bias(s, w->skip_projection_b, nullptr, h->bsp, 4, 2, 0, 0, kC));
    DSD_TRY(pack_a(s, w->final_projection_w, h->woutp, 1, 1, 32, 3, 0, 0, M, kC, kC, 1));
    DSD_TRY(pack_bias(s, w->final_projection_b, nullptr, h->boutp, 1, 3, 0, 0, M));
    HIP_TRY(hipMemcpyAsync(h->mlp0_w, w->mlp0_w, (size_t)4 * kC * kC * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->mlp0_b, w->mlp0_b, (size_t)4 * kC * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->mlp2_w, w->mlp2_w, (size_t)4 * kC * kC * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->mlp2_b, w->mlp2_b, (size_t)kC * 4, hipMemcpyDeviceToDevice, s));
    h->has_weights = true;
    if (h->split_mode) { split_kernel_attrs(); DSD_TRY(pack_split_planes(h, s)); }
    // the step table depends on the weights: rebuild for the range already known
    const int n = std::max(h->n_table, std::max(h->n_sched, 64));
    if (h->ds_table) { h->bytes -= (int64_t)h->n_table * h->L * kC * 4; dev_free(h->ds_table); }
    h->n_table = 0;
    DSD_TRY(build_step_table(h, n, s));
    h->prepared = false;    // cached conditioner projections were made with the old weights
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// schedule
// ------------------------------------------------------------------------------------------------------------
extern "C" int dsd_set_schedule(dsd_handle* h, const double* betas, int32_t n) {
    if (!h || !betas || n < 1) return fail(DSD_ERR_INVALID, "dsd_set_schedule: bad argument");
    // float64 arithmetic, then cast: shallow_diffusion_tts.py:87-123
    std::vector<double> al(n), ac(n), acp(n);
    double run = 1.0;
    for (int i = 0; i < n; ++i) {
        al[i] = 1.0 - betas[i];
        run *= al[i];
        ac[i] = run;
        acp[i] = i ? ac[i - 1] : 1.0;
    }
    for (auto& t : h->tab) t.assign(n, 0.f);
    for (int i = 0; i < n; ++i) {
        const double pv = betas[i] * (1.0 - acp[i]) / (1.0 - ac[i]);
        h->tab[0][i] = (float)betas[i];
        h->tab[1][i] = (float)ac[i];
        h->tab[2][i] = (float)acp[i];
        h->tab[3][i] = (float)std::sqrt(ac[i]);
        h->tab[4][i] = (float)std::sqrt(1.0 - ac[i]);
        h->tab[5][i] = (float)std::log(1.0 - ac[i]);
        h->tab[6][i] = (float)std::sqrt(1.0 / ac[i]);
        h->tab[7][i] = (float)std::sqrt(1.0 / ac[i] - 1.0);
        h->tab[8][i] = (float)pv;
        h->tab[9][i] = (float)std::log(std::max(pv, 1e-20));
        h->tab[10][i] = (float)(betas[i] * std::sqrt(acp[i]) / (1.0 - ac[i]));
        h->tab[11][i] = (float)((1.0 - acp[i]) * std::sqrt(al[i]) / (1.0 - ac[i]));
    }
    h->n_sched = n;
    h->has_schedule = true;
    drop_graphs(h);     // per-step scalars are baked into captured launches
    if (h->has_weights) {
        HIP_TRY(hipSetDevice(h->device));
        DSD_TRY(build_step_table(h, n, nullptr));
    }
    return DSD_OK;
}

extern "C" int dsd_get_schedule_table(dsd_handle* h, int32_t which, float* out, int32_t n) {
    if (!h || !out || which < 0 || which >= 12) return fail(DSD_ERR_INVALID, "dsd_get_schedule_table: bad argument");
    if (!h->has_schedule || n != h->n_sched) return fail(DSD_ERR_STATE, "dsd_get_schedule_table: schedule has %d steps", h->n_sched);
    std::memcpy(out, h->tab[which].data(), (size_t)n * sizeof(float));
    return DSD_OK;
}

extern "C" int dsd_set_spec_range(dsd_handle* h, const float* spec_min, const float* spec_max) {
    if (!h || !spec_min || !spec_max) return fail(DSD_ERR_INVALID, "dsd_set_spec_range: null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->spec_min_d) {
        DSD_TRY(dev_alloc(h, &h->spec_min_d, (size_t)kMPad));
        DSD_TRY(dev_alloc(h, &h->spec_max_d, (size_t)kMPad));
    }
    HIP_TRY(hipMemcpy(h->spec_min_d, spec_min, (size_t)h->M * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->spec_max_d, spec_max, (size_t)h->M * 4, hipMemcpyHostToDevice));
    h->has_spec = true;
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// prepare: workspace + hoisted conditioner projection
// ------------------------------------------------------------------------------------------------------------
static bool wino_applicable(const dsd_handle* h);

// cp[l] = Wc_l cond + bc_l + bd_l of the prepared batch (condT stays in the workspace), in the 32x32 fragment order every per-layer / latency
// kernel and k_loop read - or in the accumulator order of the Winograd loop (dsd_loop_wino.hpp)
static int launch_condproj(dsd_handle* h, bool wino, hipStream_t s) {
    CondProjParams p{h->condT, h->wcp, h->b1p, h->cp, h->TS, h->ntile32, h->ntiles, h->L, wino ? 1 : 0, {}};
    for (int l = 0; l < h->L; ++l) p.dil[l] = (unsigned char)h->dil[l];
    const int G = condproj_groups(p.dil, h->L, h->ntiles);
    hipLaunchKernelGGL(k_condproj, dim3((unsigned)h->ntiles, (unsigned)G), dim3(kThreads), condproj_lds(h->L, G), s, p);
    HIP_TRY(hipGetLastError());
    h->cp_wino = wino;
    return DSD_OK;
}
// called (outside any capture) in front of every consumer of cp: re-lays it when the batch was prepared for the other convolution
static int ensure_cp(dsd_handle* h, bool wino, hipStream_t s) {
    if (h->cp_wino == wino) return DSD_OK;
    // The re-layout flips HOST state (cp_wino) when it is ENQUEUED.  Under stream capture it would be recorded into the caller's graph and
    // the flag would flip at capture time: later eager calls and replays would disagree with it and a loop could read cp in the wrong
    // accumulator order without any error.  Refuse loudly: prepare / switch paths outside the capture.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(DSD_ERR_STATE, "the hoisted conditioner projection must change its layout for this path (%s order), and the stream is being captured: call "
                    "dsd_prepare / the path switch (dsd_set_loop_mode, dsd_set_conv_mode) and one eager call BEFORE hipStreamBeginCapture",
                    wino ? "Winograd-loop" : "direct");
    return launch_condproj(h, wino, s);
}

extern "C" int dsd_prepare(dsd_handle* h, int32_t B, int32_t T, const float* cond, int64_t sb, int64_t sh, int64_t st, void* stream) {
    if (!h || !cond) return fail(DSD_ERR_INVALID, "dsd_prepare: null argument");
    if (!h->has_weights) return fail(DSD_ERR_STATE, "dsd_prepare: call dsd_load_weights first");
    if (B < 1 || T < 1) return fail(DSD_ERR_INVALID, "dsd_prepare: B and T must be positive (got %d, %d)", B, T);
    DSD_TRY(check_sticky(h, "dsd_prepare"));
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int TS = (T + 31) / 32 * 32, ntile32 = TS / 32;
    const int64_t ntiles = (int64_t)B * ntile32;
    if (ntiles > (1 << 22)) return fail(DSD_ERR_INVALID, "dsd_prepare: batch too large (%lld tiles); split it", (long long)ntiles);
    const int64_t spec = (int64_t)B * h->M * T;
    if (ntiles > h->cap_frames || B > h->cap_B || spec > h->cap_spec) {
        HIP_TRY(hipStreamSynchronize(s));
        free_workspace(h);
        const size_t xcount = (size_t)ntiles * kC * 32 + 2 * kSlack;
        DSD_TRY(dev_alloc(h, &h->xa_base, xcount, true));
        DSD_TRY(dev_alloc(h, &h->xb_base, xcount, true));
        DSD_TRY(dev_alloc(h, &h->condT, (size_t)B * kC * TS, true));
        DSD_TRY(dev_alloc(h, &h->cp, (size_t)h->L * ntiles * 4096, true));
        DSD_TRY(dev_alloc(h, &h->skip, (size_t)ntiles * 2048, true));
        DSD_TRY(dev_alloc(h, &h->gbuf, (size_t)ntiles * kC * 32, true));
        DSD_TRY(dev_alloc(h, &h->xs, (size_t)spec, true));
        DSD_TRY(dev_alloc(h, &h->xtmp, (size_t)spec, true));
        for (auto& e : h->ering) DSD_TRY(dev_alloc(h, &e, (size_t)spec, true));
        DSD_TRY(dev_alloc(h, &h->t_dev, (size_t)B, true));
        DSD_TRY(dev_alloc(h, &h->coef_dev, (size_t)B * 5, true));
        HIP_TRY(hipMemsetAsync(h->xa_base, 0, xcount * 4, s));
        HIP_TRY(hipMemsetAsync(h->xb_base, 0, xcount * 4, s));
        h->xa = h->xa_base + kSlack;
        h->xb = h->xb_base + kSlack;
        h->cap_frames = ntiles; h->cap_B = B; h->cap_spec = spec;
    }
    h->B = B; h->T = T; h->TS = TS; h->ntile32 = ntile32; h->ntiles = (int)ntiles;

    hipLaunchKernelGGL(k_cond_layout, dim3(ntile32, kC / 32, B), dim3(32, 8), 0, s, cond, h->condT, kC, T, TS, sb, sh, st);
    HIP_TRY(hipGetLastError());
    h->prepared = true;
    // the hoisted conditioner projection in the accumulator order of the path this batch will take (a later switch of path re-lays it: ensure_cp)
    return launch_condproj(h, wino_applicable(h), s);
}

// ------------------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------------------
static int launch_inproj(dsd_handle* h, const float* spec, hipStream_t s) {
    InProjParams p{spec, h->xa, h->winp, h->binp, h->nk_in, h->M, h->T, h->TS, h->ntile32};
    hipLaunchKernelGGL(k_inproj, dim3(h->ntiles), dim3(kThreads), kMPad * 32 * 4, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

template <int G>
static void launch_lat(const LatParams& p, hipStream_t s) {
    const dim3 grid((unsigned)lat_grid(p.ntiles, G));
    bool wino = false;
    if constexpr (G != 16) {
        if (p.w1w) { hipLaunchKernelGGL((k_lat_conv_w<G>), grid, dim3(kThreads), kLatConvWLdsBytes, s, p); wino = true; }      // Winograd F(2,3) convolution
    }
    if (!wino) hipLaunchKernelGGL((k_lat_conv<G>), grid, dim3(kThreads), kLatConvLdsBytes, s, p);
    hipLaunchKernelGGL((k_lat_out<G>), grid, dim3(kThreads), kLatOutLdsBytes, s, p);
}

static int launch_layer(dsd_handle* h, int l, int t_uniform, const int* t_dev, hipStream_t s, unsigned long long* dbg = nullptr) {
    if (const int G = lat_g(h)) {
        // latency mode (dsd_lat.hpp): the layer as two kernels whose workgroups split the output rows of a tile G ways
        LatParams q{};
        q.x_in = (l & 1) ? h->xb : h->xa;
        q.x_out = (l & 1) ? h->xa : h->xb;
        q.gbuf = h->gbuf;
        q.w1p = h->w1p + (size_t)l * 4 * 96 * 256;
        q.w1q = h->w1q + (size_t)l * 16 * 96 * 64;
        q.w2p = h->w2p + (size_t)l * 4 * 32 * 256;
        q.b2 = h->b2raw + (size_t)l * 2 * kC;
        q.cp = h->cp + (size_t)l * h->ntiles * 4096;
        q.skip = h->skip;
        q.ds = h->ds_table + (size_t)l * kC;
        q.t_dev = t_dev; q.t_uniform = t_uniform; q.ds_tstride = h->L * kC;
        q.T = h->T; q.ntile32 = h->ntile32; q.ntiles = h->ntiles; q.dil = h->dil[l];
        q.first = (l == 0); q.last = (l == h->L - 1);
        // the Winograd form of the convolution for G = 2 / 4 / 8 (4 x 777: 71.0 ms against 81.9, 1 x 1550: 45.3 against 50.2, 1 x 1000: 31.2 against 32.0);
        // at G = 16 a wave's share is 128 short MFMAs and the direct kernel with its own packing stays ahead (24.6 ms against 25.0; profiles/r5_11_*)
        q.w1w = (h->conv_mode == 1 && h->w1w && G != 16) ? h->w1w + (size_t)l * kWnSteps * (kWnStepBytes / 16) : nullptr;
        if (G == 16) launch_lat<16>(q, s); else if (G == 8) launch_lat<8>(q, s); else if (G == 4) launch_lat<4>(q, s); else launch_lat<2>(q, s);
        HIP_TRY(hipGetLastError());
        return DSD_OK;
    }
    const int nb = layer_nb(h);
    LayerParams p{};
    p.x_in = (l & 1) ? h->xb : h->xa;
    p.x_out = (l & 1) ? h->xa : h->xb;
    p.w1p = h->w1p + (size_t)l * 4 * 96 * 256;
    p.w2p = h->w2p + (size_t)l * 4 * 32 * 256;
    p.b2 = h->b2raw + (size_t)l * 2 * kC;
    p.cp = h->cp + (size_t)l * h->ntiles * 4096;
    p.skip = h->skip;
    p.ds = h->ds_table + (size_t)l * kC;
    p.t_dev = t_dev;
    p.t_uniform = t_uniform;
    p.ds_tstride = h->L * kC;
    p.T = h->T; p.TS = h->TS; p.ntile32 = h->ntile32;
    p.tiles_per_utt = (h->ntile32 + nb - 1) / nb;
    p.dil = h->dil[l];
    p.first = (l == 0);
    p.dbg = dbg;
    const int total = p.tiles_per_utt * h->B;
    dim3 grid((unsigned)p.tiles_per_utt, (unsigned)h->B);
    p.wt_stores = 1;                                      // write-through epilogue stores (plain stores lost the A/B of round 1, profiles/r01c)
    p.xcd_q = total / 8; p.xcd_r = total % 8; grid = dim3((unsigned)total);      // XCD-aware workgroup -> tile map (profiles/r01b)
    const bool last = (l == h->L - 1);
    if (h->split_mode) {
        p.w1p = reinterpret_cast<const float4*>(h->w1s + (size_t)l * 4 * 48 * 12 * 64);
        p.w2p = reinterpret_cast<const float4*>(h->w2s + (size_t)l * 4 * 16 * 12 * 64);
        if (last) hipLaunchKernelGGL((k_layer_split<true>), grid, dim3(kThreads), kSplitLayerLdsBytes, s, p);
        else hipLaunchKernelGGL((k_layer_split<false>), grid, dim3(kThreads), kSplitLayerLdsBytes, s, p);
        HIP_TRY(hipGetLastError());
        return DSD_OK;
    }
    if (nb == 1) {
        if (last) hipLaunchKernelGGL((k_layer<1, true>), grid, dim3(kThreads), layer_lds_bytes<1>(), s, p);
        else hipLaunchKernelGGL((k_layer<1, false>), grid, dim3(kThreads), layer_lds_bytes<1>(), s, p);
    } else {
        if (last) hipLaunchKernelGGL((k_layer<2, true>), grid, dim3(kThreads), layer_lds_bytes<2>(), s, p);
        else hipLaunchKernelGGL((k_layer<2, false>), grid, dim3(kThreads), layer_lds_bytes<2>(), s, p);
    }
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// The L residual layers of one denoiser evaluation at step t (per-utterance steps: t_dev)
static int launch_stack(dsd_handle* h, int t_uniform, const int* t_dev, hipStream_t s) {
    for (int l = 0; l < h->L; ++l) DSD_TRY(launch_layer(h, l, t_uniform, t_dev, s));
    return DSD_OK;
}

static HeadParams head_base(dsd_handle* h) {
    HeadParams p{};
    p.skip = h->skip; p.wsp = h->wsp; p.bsp = h->bsp; p.bskp = h->bskp; p.woutp = h->woutp; p.boutp = h->boutp;
    p.winp = h->winp; p.binp = h->binp; p.x_next = h->xa;
    p.sqrt_L = (float)std::sqrt((double)h->L);
    p.nk_in = h->nk_in; p.M = h->M; p.T = h->T; p.TS = h->TS; p.ntile32 = h->ntile32;
    p.noise_cell = h->noise_cell; p.seed_cell = h->seed_cell;
    return p;
}

template <int MODE>
static int launch_head(dsd_handle* h, const HeadParams& p, bool fuse, hipStream_t s) {
    if (MODE != HEAD_EPS && lat_g(h) >= 8) {
        // G = 8 latency path: the head row-split like the layers (dsd_lat.hpp): skip projection on 8 workgroups per tile -> final projection +
        // sampler update on 3 of the 4 workgroups per tile of its grid (lat_grid(ntiles, 4); the fourth returns) -> next input projection on 8.  hbuf = the gate buffer (free behind the last layer), pbuf = the x buffer
        // the last layer read (the other one receives the next x)
        LatHeadParams q{};
        q.hp = p; q.hbuf = h->gbuf; q.pbuf = h->xb; q.ntiles = h->ntiles;
        hipLaunchKernelGGL(k_lat_head_a, dim3((unsigned)lat_grid(h->ntiles, 8)), dim3(kThreads), kLatHeadALdsBytes, s, q);
        hipLaunchKernelGGL((k_lat_head_b<MODE>), dim3((unsigned)lat_grid(h->ntiles, 4)), dim3(kThreads), kLatHeadBLdsBytes, s, q);
        if (fuse) hipLaunchKernelGGL(k_lat_head_c, dim3((unsigned)lat_grid(h->ntiles, 8)), dim3(kThreads), kLatHeadCLdsBytes, s, q);
        HIP_TRY(hipGetLastError());
        return DSD_OK;
    }
    if (fuse) hipLaunchKernelGGL((k_head<MODE, true>), dim3(h->ntiles), dim3(kThreads), kHeadLdsBytes, s, p);
    else hipLaunchKernelGGL((k_head<MODE, false>), dim3(h->ntiles), dim3(kThreads), kHeadLdsBytes, s, p);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static int check_ready(dsd_handle* h, const char* who, bool need_sched) {
    if (!h) return fail(DSD_ERR_INVALID, "%s: null handle", who);
    if (!h->has_weights) return fail(DSD_ERR_STATE, "%s: weights not loaded", who);
    if (!h->prepared) return fail(DSD_ERR_STATE, "%s: no batch prepared (dsd_prepare)", who);
    if (need_sched && !h->has_schedule) return fail(DSD_ERR_STATE, "%s: schedule not set (dsd_set_schedule)", who);
    return check_sticky(h, who);
}

// ------------------------------------------------------------------------------------------------------------
// single evaluation
// ------------------------------------------------------------------------------------------------------------
extern "C" int dsd_denoise(dsd_handle* h, const float* x, const int32_t* t, float* eps, void* stream) {
    DSD_TRY(check_ready(h, "dsd_denoise", false));
    if (!x || !t || !eps) return fail(DSD_ERR_INVALID, "dsd_denoise: null argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int tmax = 0;
    bool uniform = true;
    for (int b = 0; b < h->B; ++b) {
        if (t[b] < 0) return fail(DSD_ERR_INVALID, "dsd_denoise: negative step index");
        tmax = std::max(tmax, (int)t[b]);
        uniform = uniform && (t[b] == t[0]);
    }
    DSD_TRY(build_step_table(h, tmax + 1, s));
    const int* t_dev = nullptr;
    if (!uniform) {
        // t is a caller-owned host array: it is copied into a pinned staging slot here, so it is not referenced after return and
        // the H2D copy is asynchronous - no stream synchronisation
        char* pin = nullptr; int slot = 0;
        DSD_TRY(pin_acquire(h, (size_t)h->B * 4, &pin, &slot));
        std::memcpy(pin, t, (size_t)h->B * 4);
        HIP_TRY(hipMemcpyAsync(h->t_dev, pin, (size_t)h->B * 4, hipMemcpyHostToDevice, s));
        DSD_TRY(pin_release(h, slot, s));
        t_dev = h->t_dev;
    }
    DSD_TRY(ensure_cp(h, false, s));
    DSD_TRY(launch_inproj(h, x, s));
    DSD_TRY(launch_stack(h, t[0], t_dev, s));
    HeadParams p = head_base(h);
    p.eps_out = eps;
    return launch_head<HEAD_EPS>(h, p, false, s);
}

extern "C" int dsd_q_sample(dsd_handle* h, const float* x_start, const float* noise, int32_t t, float* out, void* stream) {
    if (!h || !x_start || !noise || !out) return fail(DSD_ERR_INVALID, "dsd_q_sample: null argument");
    if (!h->has_schedule) return fail(DSD_ERR_STATE, "dsd_q_sample: schedule not set");
    if (!h->prepared) return fail(DSD_ERR_STATE, "dsd_q_sample: no batch prepared");
    if (t < 0 || t >= h->n_sched) return fail(DSD_ERR_INVALID, "dsd_q_sample: t=%d outside the %d-step schedule", t, h->n_sched);
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->B * h->M * h->T;
    hipLaunchKernelGGL(k_qsample, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       x_start, noise, out, h->tab[3][t], h->tab[4][t], n);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsd_norm_spec(dsd_handle* h, const float* mel, float* x, void* stream) {
    if (!h || !mel || !x) return fail(DSD_ERR_INVALID, "dsd_norm_spec: null argument");
    if (!h->has_spec) return fail(DSD_ERR_STATE, "dsd_norm_spec: spec range not set");
    if (!h->prepared) return fail(DSD_ERR_STATE, "dsd_norm_spec: no batch prepared");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_norm_spec, dim3((h->T + 31) / 32, h->B), dim3(256), 32 * (h->M + 1) * 4, (hipStream_t)stream, mel, x,
                       h->spec_min_d, h->spec_max_d, h->M, h->T);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsd_denorm_spec(dsd_handle* h, const float* x, const float* mask, float* mel, void* stream) {
    if (!h || !mel || !x) return fail(DSD_ERR_INVALID, "dsd_denorm_spec: null argument");
    if (!h->has_spec) return fail(DSD_ERR_STATE, "dsd_denorm_spec: spec range not set");
    if (!h->prepared) return fail(DSD_ERR_STATE, "dsd_denorm_spec: no batch prepared");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_denorm_spec, dim3((h->T + 31) / 32, h->B), dim3(256), 32 * (h->M + 1) * 4, (hipStream_t)stream, x, mask, mel,
                       h->spec_min_d, h->spec_max_d, h->M, h->T);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// sampling loops
// ------------------------------------------------------------------------------------------------------------
// Enqueue the whole DDPM loop on stream s, operating on the internal spec buffer h->xs.
static int enqueue_ddpm(dsd_handle* h, int k_step, hipStream_t s) {
    const size_t bmt = (size_t)h->B * h->M * h->T;
    DSD_TRY(launch_inproj(h, h->xs, s));
    for (int j = 0; j < k_step; ++j) {
        const int t = k_step - 1 - j;
        DSD_TRY(launch_stack(h, t, nullptr, s));
        HeadParams p = head_base(h);
        p.x_base = h->xs; p.x_out = h->xs;
        p.noise_off = (size_t)j * bmt; p.step_id = (unsigned)j;
        p.sa = h->tab[6][t]; p.sb = h->tab[7][t]; p.c1 = h->tab[10][t]; p.c2 = h->tab[11][t];
        // nonzero_mask * exp(0.5 * logvar): fp32 like the reference's [B,1,1,1] tensors (:165-166)
        p.sigma = (t == 0) ? 0.f : std::exp(0.5f * h->tab[9][t]);
        DSD_TRY(launch_head<HEAD_DDPM>(h, p, /*fuse next in-proj*/ t > 0, s));
    }
    return DSD_OK;
}

// get_x_pred coefficients (shallow_diffusion_tts.py:174-185), in fp32 like the reference
static void plms_coef(const dsd_handle* h, int t, int interval, float* dA, float* cx, float* ce) {
    const float a_t = h->tab[1][t];
    const float a_prev = (t < interval) ? 1.0f : h->tab[1][std::max(t - interval, 0)];
    const float a_t_sq = std::sqrt(a_t), a_prev_sq = std::sqrt(a_prev);
    *dA = a_prev - a_t;
    *cx = 1.0f / (a_t_sq * (a_t_sq + a_prev_sq));
    *ce = 1.0f / (a_t_sq * (std::sqrt((1.0f - a_prev) * a_t) + std::sqrt((1.0f - a_t) * a_prev)));
}

static int enqueue_plms(dsd_handle* h, int k_step, int interval, hipStream_t s) {
    DSD_TRY(launch_inproj(h, h->xs, s));
    int hist = 0;           // len(noise_list), capped at 3 for the formula choice
    int head_slot = 0;      // ring slot receiving the next stored eps
    std::vector<int> ts;
    for (int i = 0; i < k_step; i += interval) ts.push_back(i);
    for (int n = (int)ts.size() - 1; n >= 0; --n) {
        const int t = ts[n];
        const bool more = n > 0;
        float dA, cx, ce;
        plms_coef(h, t, interval, &dA, &cx, &ce);
        DSD_TRY(launch_stack(h, t, nullptr, s));
        HeadParams p = head_base(h);
        p.dA = dA; p.cx = cx; p.ce = ce;
        float* slot = h->ering[head_slot & 3];
        if (hist == 0) {
            // warm-up (:188-192): x_pred from the raw eps, second evaluation at max(t - interval, 0)
            p.order = PLMS_RAW; p.x_base = h->xs; p.x_out = h->xtmp; p.eps_out = slot;
            DSD_TRY(launch_head<HEAD_PLMS>(h, p, true, s));
            const int t_prev = std::max(t - interval, 0);
            DSD_TRY(launch_stack(h, t_prev, nullptr, s));
            HeadParams q = head_base(h);
            q.dA = dA; q.cx = cx; q.ce = ce;
            q.order = PLMS_HEUN; q.x_base = h->xs; q.x_out = h->xs; q.eps_out = nullptr; q.e1 = slot;
            DSD_TRY(launch_head<HEAD_PLMS>(h, q, more, s));
        } else {
            p.order = (hist == 1) ? PLMS_AB2 : (hist == 2) ? PLMS_AB3 : PLMS_AB4;
            p.x_base = h->xs; p.x_out = h->xs; p.eps_out = slot;
            p.e1 = h->ering[(head_slot - 1) & 3]; p.e2 = h->ering[(head_slot - 2) & 3]; p.e3 = h->ering[(head_slot - 3) & 3];
            DSD_TRY(launch_head<HEAD_PLMS>(h, p, more, s));
        }
        ++head_slot;
        hist = std::min(hist + 1, 3);
    }
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------------------------
// persistent loop (dsd_loop.hpp)
// ------------------------------------------------------------------------------------------------------------
// The per-evaluation head parameters, exactly what enqueue_ddpm / enqueue_plms bake into their launches.
static void plan_evals(dsd_handle* h, int kind, int k_step, int interval, std::vector<HeadParams>& ev, std::vector<int>& ts_out) {
    const size_t bmt = (size_t)h->B * h->M * h->T;
    if (kind == 0) {
        for (int j = 0; j < k_step; ++j) {
            const int t = k_step - 1 - j;
            HeadParams p = head_base(h);
            p.x_base = h->xs; p.x_out = h->xs;
            p.noise_off = (size_t)j * bmt; p.step_id = (unsigned)j;
            p.sa = h->tab[6][t]; p.sb = h->tab[7][t]; p.c1 = h->tab[10][t]; p.c2 = h->tab[11][t];
            p.sigma = (t == 0) ? 0.f : std::exp(0.5f * h->tab[9][t]);
            ev.push_back(p); ts_out.push_back(t);
        }
        return;
    }
    int hist = 0, head_slot = 0;
    std::vector<int> ts;
    for (int i = 0; i < k_step; i += interval) ts.push_back(i);
    for (int n = (int)ts.size() - 1; n >= 0; --n) {
        const int t = ts[n];
        float dA, cx, ce;
        plms_coef(h, t, interval, &dA, &cx, &ce);
        HeadParams p = head_base(h);
        p.dA = dA; p.cx = cx; p.ce = ce;
        float* slot = h->ering[head_slot & 3];
        if (hist == 0) {
            p.order = PLMS_RAW; p.x_base = h->xs; p.x_out = h->xtmp; p.eps_out = slot;
            ev.push_back(p); ts_out.push_back(t);
            HeadParams q = head_base(h);
            q.dA = dA; q.cx = cx; q.ce = ce;
            q.order = PLMS_HEUN; q.x_base = h->xs; q.x_out = h->xs; q.eps_out = nullptr; q.e1 = slot;
            ev.push_back(q); ts_out.push_back(std::max(t - interval, 0));
        } else {
            p.order = (hist == 1) ? PLMS_AB2 : (hist == 2) ? PLMS_AB3 : PLMS_AB4;
            p.x_base = h->xs; p.x_out = h->xs; p.eps_out = slot;
            p.e1 = h->ering[(head_slot - 1) & 3]; p.e2 = h->ering[(head_slot - 2) & 3]; p.e3 = h->ering[(head_slot - 3) & 3];
            ev.push_back(p); ts_out.push_back(t);
        }
        ++head_slot;
        hist = std::min(hist + 1, 3);
    }
}

// ONE persistent loop at a time per device: all workgroups of a k_loop launch wait for each other, so two such launches on two
// streams of one GPU (two handles = two models of one process, or one handle driven from two streams) could each hold part of the
// CUs and starve the other into its timeout.  Every persistent launch therefore waits (on the device, hipStreamWaitEvent) for the
// previous persistent launch of this process on the same device when that one went to a DIFFERENT stream, and records the event
// the next one will wait for.  The loop fills the chip anyway, so the serialisation costs nothing.
static const int kMaxDevices = 64;
static std::mutex g_loop_mu[kMaxDevices];
static hipEvent_t g_loop_ev[kMaxDevices];
static hipStream_t g_loop_stream[kMaxDevices];
static bool g_loop_has[kMaxDevices];

// true when the prepared batch can run as the persistent loop: 32-frame tiles, a whole utterance fits the co-resident grid
static bool loop_applicable(const dsd_handle* h) {
    if (!((h->loop_mode == 1 || h->loop_mode == 2) && !h->persist_off && h->use_graph && layer_nb(h) == 1 && h->n_cu >= 8 &&
          h->ntile32 <= h->n_cu && h->L <= kLoopMaxLayers)) return false;
    if (h->loop_mode == 2) {
        if (lat_g(h)) return false;
        // chunks of whole utterances may leave much of the chip idle (T = 5000: 157 tiles per launch on 256 CUs); the per-layer kernels
        // have no such constraint, only the wave quantisation of their grid, and cost ~5 % more at equal occupancy
        const int upc = std::max(1, h->n_cu / h->ntile32), chunks = (h->B + upc - 1) / upc;
        const double u_p = (double)h->ntiles / ((double)chunks * h->n_cu);
        // (the per-layer kernels evaluate the direct convolution: at equal occupancy they take 1.05 x the direct loop's time and 1.29 x the
        // Winograd loop's - 133 ms against 127 / 103.7 ms per 256 tiles)
        const double rel = (h->conv_mode == 1 && h->w1w && !h->split_mode) ? 0.78 : 0.95;
        const double u_l = rel * (double)h->ntiles / ((double)((h->ntiles + h->n_cu - 1) / h->n_cu) * h->n_cu);
        if (u_l > u_p) return false;
    }
    return true;
}

// the prepared batch takes the persistent loop AND that loop evaluates the dilated convolution as Winograd F(2,3) (dsd_loop_wino.hpp)
static bool wino_applicable(const dsd_handle* h) { return h->conv_mode == 1 && !h->split_mode && h->w1w && loop_applicable(h); }

static int get_plan(dsd_handle* h, int kind, int k_step, int interval, dsd_handle::LoopPlan** out);

static int run_persistent(dsd_handle* h, int kind, int k_step, int interval, hipStream_t s) {
    dsd_handle::LoopPlan* plan = nullptr;
    DSD_TRY(get_plan(h, kind, k_step, interval, &plan));
    if (h->loop_cap_tiles < h->ntiles) {
        dev_free(h->loop_flags); dev_free(h->loop_halo);
        h->loop_tmo_at = -1;
        DSD_TRY(dev_alloc(h, &h->loop_flags, (size_t)h->ntiles + 64, true));
        DSD_TRY(dev_alloc(h, &h->loop_halo, (size_t)2 * h->ntiles * 2 * kC * 8, true));
        h->loop_cap_tiles = h->ntiles;
    }
    HIP_TRY(hipMemsetAsync(h->loop_flags, 0, ((size_t)h->ntiles + 64) * sizeof(unsigned), s));
    const bool wino = wino_applicable(h);
    DSD_TRY(ensure_cp(h, wino, s));
    LoopParams p{};
    p.w1p = h->w1p; p.w2p = h->w2p; p.b2raw = h->b2raw; p.cp = h->cp; p.cp_lstride = (size_t)h->ntiles * 4096;
    p.ds_table = h->ds_table;
    p.L = h->L; p.T = h->T; p.TS = h->TS; p.ntile32 = h->ntile32; p.ntiles_total = h->ntiles;
    for (int l = 0; l < h->L; ++l) p.dil[l] = (unsigned char)h->dil[l];
    p.head = head_base(h);
    p.evals = plan->evals; p.eval_t = plan->eval_t; p.n_evals = plan->n_evals;
    p.spec0 = h->xs;
    p.flags = h->loop_flags; p.halo = h->loop_halo; p.tmo = h->loop_flags + h->ntiles;
    h->loop_tmo_at = h->ntiles;
    p.dbg = h->loop_dbg; p.dbg_phase = h->loop_dbg_phase;
    // chunks of whole utterances, at most one workgroup per CU (all workgroups of a launch wait for each other)
    const int utt_per_chunk = std::max(1, h->n_cu / h->ntile32);
    const int dv = (h->device >= 0 && h->device < kMaxDevices) ? h->device : 0;
    std::lock_guard<std::mutex> guard(g_loop_mu[dv]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool guarded = (cap == hipStreamCaptureStatusNone);
    if (guarded) {
        if (!g_loop_ev[dv]) HIP_TRY(hipEventCreateWithFlags(&g_loop_ev[dv], hipEventDisableTiming));
        if (g_loop_has[dv] && g_loop_stream[dv] != s) HIP_TRY(hipStreamWaitEvent(s, g_loop_ev[dv], 0));
    }
    for (int b0 = 0; b0 < h->B; b0 += utt_per_chunk) {
        const int nb = std::min(utt_per_chunk, h->B - b0);
        p.tile_base = b0 * h->ntile32; p.n_tiles = nb * h->ntile32;
        if (h->split_mode) {
            // EXPERIMENT (dsd_loop_split.hpp): the same loop with the layers' contractions as six bf16 plane products per fp32 product
            const LoopSplitParams q = h->split_w == 2 ? LoopSplitParams{p, h->wl2, h->wl2 + (size_t)48 * 4 * 512, (unsigned)((size_t)h->L * 64 * 4 * 8192), h->split_touch}
                                                        : LoopSplitParams{p, h->wlc, h->wlc + (size_t)48 * 4 * 768, (unsigned)((size_t)h->L * 64 * 4 * 12288), h->split_touch};
            const dim3 grid((unsigned)p.n_tiles), block(kThreads);
#define DSD_LAUNCH_SPLIT(WF) do { if (kind == 0) hipLaunchKernelGGL((k_loop_split<HEAD_DDPM, WF>), grid, block, kLoopSplitLdsBytes, s, q); \
                                  else hipLaunchKernelGGL((k_loop_split<HEAD_PLMS, WF>), grid, block, kLoopSplitLdsBytes, s, q); } while (0)
            if (h->split_w == 2) DSD_LAUNCH_SPLIT(2);
            else DSD_LAUNCH_SPLIT(0);
#undef DSD_LAUNCH_SPLIT
        } else if (wino) {
            // Winograd F(2,3) form of the dilated convolution (dsd_loop_wino.hpp): the default of this path
            const LoopWinoParams q{p, h->w1w, (unsigned)((size_t)h->L * kWnSteps * kWnStepBytes), h->wino_touch};
            const dim3 grid((unsigned)p.n_tiles), block(kThreads);
            if (kind == 0) hipLaunchKernelGGL((k_loop_wino<HEAD_DDPM, 4>), grid, block, kLoopWinoLdsBytes, s, q);
            else hipLaunchKernelGGL((k_loop_wino<HEAD_PLMS, 4>), grid, block, kLoopWinoLdsBytes, s, q);
        } else if (kind == 0) hipLaunchKernelGGL((k_loop<HEAD_DDPM>), dim3((unsigned)p.n_tiles), dim3(kThreads), kLoopLdsBytes, s, p);
        else hipLaunchKernelGGL((k_loop<HEAD_PLMS>), dim3((unsigned)p.n_tiles), dim3(kThreads), kLoopLdsBytes, s, p);
        HIP_TRY(hipGetLastError());
    }
    DSD_TRY(sticky_alloc(h));
    hipLaunchKernelGGL(k_latch_tmo, dim3(1), dim3(1), 0, s, (const unsigned*)p.tmo, h->sticky_dev);
    HIP_TRY(hipGetLastError());
    if (guarded) {
        HIP_TRY(hipEventRecord(g_loop_ev[dv], s));
        g_loop_stream[dv] = s;
        g_loop_has[dv] = true;
    }
    return DSD_OK;
}

static int get_plan(dsd_handle* h, int kind, int k_step, int interval, dsd_handle::LoopPlan** out) {
    const GraphKey key{kind, h->B, h->T, k_step, interval, 32};
    auto it = h->plans.find(key);
    if (it == h->plans.end()) {
        std::vector<HeadParams> ev; std::vector<int> ts;
        plan_evals(h, kind, k_step, interval, ev, ts);
        dsd_handle::LoopPlan pl;
        pl.n_evals = (int)ev.size();
        HIP_TRY(hipMalloc((void**)&pl.evals, ev.size() * sizeof(HeadParams)));
        HIP_TRY(hipMalloc((void**)&pl.eval_t, ts.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(pl.evals, ev.data(), ev.size() * sizeof(HeadParams), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(pl.eval_t, ts.data(), ts.size() * sizeof(int), hipMemcpyHostToDevice));
        if (h->plans.size() >= 8) drop_graphs(h);
        it = h->plans.emplace(key, pl).first;
    }
    *out = &it->second;
    return DSD_OK;
}

static int run_loop(dsd_handle* h, int kind, float* x, const float* noise, int k_step, int interval, hipStream_t s) {
    const size_t bmt = (size_t)h->B * h->M * h->T;
    HIP_TRY(hipMemcpyAsync(h->xs, x, bmt * 4, hipMemcpyDeviceToDevice, s));
    if (kind == 0) {
        hipLaunchKernelGGL(k_set_cell, dim3(1), dim3(1), 0, s, h->noise_cell, noise);       // nullptr selects the Philox draws
        if (!noise) hipLaunchKernelGGL(k_set_seed, dim3(1), dim3(1), 0, s, h->seed_cell, h->noise_seed);
        HIP_TRY(hipGetLastError());
    }
    if (loop_applicable(h)) {
        DSD_TRY(run_persistent(h, kind, k_step, interval, s));
    } else if (!h->use_graph) {
        DSD_TRY(ensure_cp(h, false, s));
        DSD_TRY(kind == 0 ? enqueue_ddpm(h, k_step, s) : enqueue_plms(h, k_step, interval, s));
    } else {
        DSD_TRY(ensure_cp(h, false, s));
        const GraphKey key{kind, h->B, h->T, k_step, interval, layer_nb(h) + 100 * lat_g(h) + 10000 * ((h->conv_mode == 1 && h->w1w) ? 1 : 0)};      // (the latency nodes differ by convolution form)
        auto it = h->graphs.find(key);
        if (it == h->graphs.end()) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
            const int rc = (kind == 0) ? enqueue_ddpm(h, k_step, h->cap_stream) : enqueue_plms(h, k_step, interval, h->cap_stream);
            const hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
            if (rc != DSD_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return fail(DSD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            hipGraphExec_t ge = nullptr;
            const hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e2 != hipSuccess) return fail(DSD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
            if (h->graphs.size() >= 8) drop_graphs(h);
            it = h->graphs.emplace(key, ge).first;
        }
        HIP_TRY(hipGraphLaunch(it->second, s));
    }
    if (h->persist_off && ++h->parked_calls >= kParkedCalls) { h->persist_off = false; h->parked_calls = 0; }     // re-arm the persistent path
    HIP_TRY(hipMemcpyAsync(x, h->xs, bmt * 4, hipMemcpyDeviceToDevice, s));
    return DSD_OK;
}

extern "C" int dsd_set_noise_seed(dsd_handle* h, uint64_t seed) {
    if (!h) return fail(DSD_ERR_INVALID, "dsd_set_noise_seed: null handle");
    h->noise_seed = seed;
    return DSD_OK;
}

extern "C" int dsd_philox_normal(dsd_handle* h, uint64_t seed, int32_t step, float* out, int64_t n, void* stream) {
    if (!h || !out || n < 1 || step < 0) return fail(DSD_ERR_INVALID, "dsd_philox_normal: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_philox_fill, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, out, (size_t)n,
                       (unsigned long long)seed, (unsigned)step);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

static void split_kernel_attrs() {
    if (first_on_device(1)) {
        (void)hipFuncSetAttribute((const void*)k_layer_split<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLayerLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_layer_split<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLayerLdsBytes);
#define DSD_SPLIT_ATTR(WF) do { \
            (void)hipFuncSetAttribute((const void*)k_loop_split<HEAD_DDPM, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopSplitLdsBytes); \
            (void)hipFuncSetAttribute((const void*)k_loop_split<HEAD_PLMS, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopSplitLdsBytes); } while (0)
        DSD_SPLIT_ATTR(0); DSD_SPLIT_ATTR(2);
#undef DSD_SPLIT_ATTR
    }
}

// Weight planes of the split-precision kernels, derived on the device from the fp32 fragment-order weights: the bf16 planes of the per-layer
// kernel (w1s / w2s) and ONE stream for the persistent split loop - the format split_w selects (the pair format, or the bf16 planes in
// consumption order); the other one is packed only if the format is switched (DSD_SPLIT_W at dsd_create).
static int pack_split_planes(dsd_handle* h, hipStream_t s) {
    const int L = h->L;
    if (!h->w1s) {
        DSD_TRY(dev_alloc(h, &h->w1s, (size_t)L * 4 * 48 * 12 * 64 + kWeightSlack));
        DSD_TRY(dev_alloc(h, &h->w2s, (size_t)L * 4 * 16 * 12 * 64 + kWeightSlack));
        HIP_TRY(hipMemsetAsync(h->w1s + (size_t)L * 4 * 48 * 12 * 64, 0, (size_t)kWeightSlack * 16, s));
        HIP_TRY(hipMemsetAsync(h->w2s + (size_t)L * 4 * 16 * 12 * 64, 0, (size_t)kWeightSlack * 16, s));
    }
    if (h->split_w == 2 && !h->wl2) {
        DSD_TRY(dev_alloc(h, &h->wl2, (size_t)L * 64 * 4 * 512 + kWeightSlack));
        HIP_TRY(hipMemsetAsync(h->wl2 + (size_t)L * 64 * 4 * 512, 0, (size_t)kWeightSlack * 16, s));
    }
    if (h->split_w == 0 && !h->wlc) {
        DSD_TRY(dev_alloc(h, &h->wlc, (size_t)L * 64 * 4 * 768 + kWeightSlack));
        HIP_TRY(hipMemsetAsync(h->wlc + (size_t)L * 64 * 4 * 768, 0, (size_t)kWeightSlack * 16, s));
    }
    for (int l = 0; l < L; ++l) {
        const float* w1 = reinterpret_cast<const float*>(h->w1p + (size_t)l * 4 * 96 * 256);
        const float* w2 = reinterpret_cast<const float*>(h->w2p + (size_t)l * 4 * 32 * 256);
        hipLaunchKernelGGL((k_pack_split<false>), dim3(1024), dim3(256), 0, s, w1, reinterpret_cast<su16*>(h->w1s + (size_t)l * 4 * 48 * 12 * 64), 4, 16, 3, 0, 0LL, 0LL);
        hipLaunchKernelGGL((k_pack_split<false>), dim3(512), dim3(256), 0, s, w2, reinterpret_cast<su16*>(h->w2s + (size_t)l * 4 * 16 * 12 * 64), 4, 16, 1, 0, 0LL, 0LL);
        if (h->split_w == 0) {
            su16* lc = reinterpret_cast<su16*>(h->wlc + (size_t)l * 64 * 4 * 768);       // consumption order: chunk stride 4 x 6144, wave stride 6144
            hipLaunchKernelGGL((k_pack_split<false>), dim3(1024), dim3(256), 0, s, w1, lc, 4, 16, 3, 1, 6144LL, 4 * 6144LL);
            hipLaunchKernelGGL((k_pack_split<false>), dim3(512), dim3(256), 0, s, w2, lc + (size_t)48 * 4 * 6144, 4, 16, 1, 0, 6144LL, 4 * 6144LL);
        } else {
            su16* l2 = reinterpret_cast<su16*>(h->wl2 + (size_t)l * 64 * 4 * 512);       // pair format: chunk stride 4 x 4096, wave stride 4096
            hipLaunchKernelGGL((k_pack_split<true>), dim3(1024), dim3(256), 0, s, w1, l2, 4, 16, 3, 1, 4096LL, 4 * 4096LL);
            hipLaunchKernelGGL((k_pack_split<true>), dim3(512), dim3(256), 0, s, w2, l2 + (size_t)48 * 4 * 4096, 4, 16, 1, 0, 4096LL, 4 * 4096LL);
        }
    }
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsd_set_split_mode(dsd_handle* h, int32_t on, void* stream) {
    if (!h || on < 0 || on > 1) return fail(DSD_ERR_INVALID, "dsd_set_split_mode: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (on && !h->split_mode) {
        split_kernel_attrs();
        if (h->has_weights) DSD_TRY(pack_split_planes(h, (hipStream_t)stream));
    }
    if ((on != 0) != h->split_mode) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        drop_graphs(h);                                   // captured graphs hold the other layer kernel
    }
    h->split_mode = on != 0;
    return DSD_OK;
}

extern "C" int dsd_get_split_mode(dsd_handle* h) { return (h && h->split_mode) ? 1 : 0; }

// Debug hook: ONE residual layer (the fp32 kernel, or the split-precision one while that mode is on) on a caller-supplied input, results
// in logical layout - for layer-level parity tests and error localisation.  x_in / x_out / skip_out: DEVICE [B][C][TS] (TS = T up to 32;
// x_out is not written for the last layer).  Uses the prepared batch's cp; overwrites the handle's x and skip work buffers.
extern "C" int dsd_debug_layer(dsd_handle* h, int32_t layer, int32_t t, const float* x_in, float* x_out, float* skip_out, void* stream) {
    DSD_TRY(check_ready(h, "dsd_debug_layer", false));
    if (!x_in || !skip_out || layer < 0 || layer >= h->L || t < 0 || (!x_out && layer != h->L - 1))
        return fail(DSD_ERR_INVALID, "dsd_debug_layer: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DSD_TRY(build_step_table(h, t + 1, s));
    DSD_TRY(ensure_cp(h, false, s));
    float* xin = (layer & 1) ? h->xb : h->xa;
    float* xout = (layer & 1) ? h->xa : h->xb;
    hipLaunchKernelGGL(k_dbg_to_tiles, dim3((unsigned)h->ntiles), dim3(256), 0, s, const_cast<float*>(x_in), xin, h->TS, h->ntile32, 1);
    HIP_TRY(hipMemsetAsync(h->skip, 0, (size_t)h->ntiles * 2048 * sizeof(float4), s));
    DSD_TRY(launch_layer(h, layer, t, nullptr, s));
    if (x_out && layer != h->L - 1)
        hipLaunchKernelGGL(k_dbg_to_tiles, dim3((unsigned)h->ntiles), dim3(256), 0, s, x_out, xout, h->TS, h->ntile32, 0);
    hipLaunchKernelGGL(k_dbg_skip_to_logical, dim3((unsigned)h->ntiles), dim3(256), 0, s, h->skip, skip_out, h->TS, h->ntile32);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsd_set_loop_mode(dsd_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 3)
        return fail(DSD_ERR_INVALID, "dsd_set_loop_mode: mode must be 0 (per-layer kernels), 1 (persistent loop), 2 (automatic) or 3 (latency kernels)");
    h->loop_mode = mode;
    h->persist_off = false;           // an explicit choice re-arms the persistent path after a reported timeout
    h->parked_calls = 0;
    return DSD_OK;
}

extern "C" int dsd_loop_parked(dsd_handle* h) { return (h && h->persist_off) ? kParkedCalls - h->parked_calls : 0; }

extern "C" int dsd_set_lat_split(dsd_handle* h, int32_t g) {
    if (!h || !(g == -1 || g == 0 || g == 2 || g == 4 || g == 8 || g == 16))
        return fail(DSD_ERR_INVALID, "dsd_set_lat_split: g must be -1 (by batch size), 0, 2, 4, 8 or 16");
    h->lat_req = g;
    return DSD_OK;
}

extern "C" int dsd_get_lat_split(dsd_handle* h) { return (h && h->prepared) ? lat_g(h) : 0; }

extern "C" int dsd_set_conv_mode(dsd_handle* h, int32_t mode, int32_t touch_ahead) {
    if (!h || mode < 0 || mode > 1) return fail(DSD_ERR_INVALID, "dsd_set_conv_mode: mode must be 0 (direct K = 768 contraction) or 1 (Winograd F(2,3))");
    if (touch_ahead < -1 || touch_ahead > 64) return fail(DSD_ERR_INVALID, "dsd_set_conv_mode: touch_ahead must be -1 (keep), 0 (off) .. 64 steps");
    h->conv_mode = mode;
    if (touch_ahead >= 0) h->wino_touch = touch_ahead;
    return DSD_OK;
}

// 1 when the dilated convolution of the prepared batch runs as Winograd F(2,3): the persistent loop k_loop_wino, or the conv node of the
// row-split latency kernels at G = 2 / 4 / 8 (k_lat_conv_w, launch_layer); 0: the direct form (mode 0, G = 16, per-layer kernels, split mode)
extern "C" int dsd_get_conv_mode(dsd_handle* h) {
    if (!h || !h->prepared) return 0;
    if (wino_applicable(h)) return 1;
    const int g = loop_applicable(h) ? 0 : lat_g(h);
    return (h->conv_mode == 1 && h->w1w && !h->split_mode && (g == 2 || g == 4 || g == 8)) ? 1 : 0;
}

extern "C" int dsd_get_loop_mode(dsd_handle* h) { return (h && h->prepared && loop_applicable(h)) ? 1 : 0; }

extern "C" int dsd_loop_launches(dsd_handle* h) {
    if (!h || !h->prepared || !loop_applicable(h)) return 0;
    const int utt_per_chunk = std::max(1, h->n_cu / h->ntile32);
    return (h->B + utt_per_chunk - 1) / utt_per_chunk;
}

extern "C" int dsd_loop_timeouts(dsd_handle* h, void* stream) {
    if (!h) return fail(DSD_ERR_INVALID, "dsd_loop_timeouts: null handle");
    if (!h->loop_flags || h->loop_tmo_at < 0) return 0;
    unsigned v = 0;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    // the word of the last persistent run (a batch prepared SINCE then may have fewer tiles: flags[ntiles] would be one of that run's tile flags)
    HIP_TRY(hipMemcpy(&v, h->loop_flags + h->loop_tmo_at, sizeof v, hipMemcpyDeviceToHost));
    return (int)v;
}

extern "C" int dsd_check(dsd_handle* h) {
    if (!h) return fail(DSD_ERR_INVALID, "dsd_check: null handle");
    return check_sticky(h, "dsd_check");
}

extern "C" int32_t dsd_debug_condproj_groups(const uint8_t* dilations, int32_t L, int32_t ntiles, int64_t* lds_bytes) {
    if (!dilations || L < 1 || L > 64 || ntiles < 1) return -1;
    const int G = condproj_groups(dilations, L, ntiles);
    if (lds_bytes) *lds_bytes = (int64_t)condproj_lds(L, G);
    return G;
}

extern "C" int dsd_debug_hold_cus(int32_t device, int32_t n_workgroups, int32_t milliseconds, uint32_t* started, void* stream) {
    if (n_workgroups < 1 || n_workgroups > 4096 || milliseconds < 1 || milliseconds > 300000)
        return fail(DSD_ERR_INVALID, "dsd_debug_hold_cus: 1..4096 workgroups, 1..300000 ms");
    HIP_TRY(hipSetDevice(device));
    const int lds = 160 * 1024;                                    // the whole LDS of a CU: one holder per CU, nothing else fits beside it
    if (first_on_device(2)) HIP_TRY(hipFuncSetAttribute((const void*)k_hold_cu, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k_hold_cu, dim3((unsigned)n_workgroups), dim3(64), lds, (hipStream_t)stream, (unsigned long long)milliseconds * 100000ull,
                       (unsigned*)started);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

// Debug hook: run the persistent DDPM loop once on the prepared batch (x, noise as for dsd_sample_ddpm) with per-wave shader-clock
// stamps taken in phase `phase` (= evaluation * L + layer; pick a non-last layer): HOST out[n_wg * 4 * 16] u64 with, per wave,
// [0..7] {phase start, neighbours' flags seen, y tile staged, conv done, gate done, x' ready, halo published, phase end} and
// [8..15] the HEAD of that evaluation {last layer done, skip tile scaled + staged, skip projection done, its ReLU tile visible, final
// projection done (waves 0-2), sampler update stored, barrier, next input projection + halo published}.
extern "C" int dsd_debug_loop_timeline(dsd_handle* h, float* x, const float* noise, int32_t k_step, int32_t phase, uint64_t* out,
                                       int32_t max_wg, int32_t* n_wg, void* stream) {
    DSD_TRY(check_ready(h, "dsd_debug_loop_timeline", true));
    if (!x || !noise || !out || !n_wg) return fail(DSD_ERR_INVALID, "dsd_debug_loop_timeline: null argument");
    if (!loop_applicable(h)) return fail(DSD_ERR_STATE, "dsd_debug_loop_timeline: the prepared batch does not take a persistent path");
    if (k_step < 2 || phase / h->L >= k_step - 1) return fail(DSD_ERR_INVALID, "dsd_debug_loop_timeline: pick a phase of an evaluation that is not the last");
    const int nwg = h->ntiles;                   // one stamp block per workgroup = per tile
    if (h->ntiles > h->n_cu || nwg > max_wg) return fail(DSD_ERR_INVALID, "dsd_debug_loop_timeline: needs a single-launch batch (%d workgroups)", nwg);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DSD_TRY(build_step_table(h, h->n_sched, s));
    HIP_TRY(hipMalloc((void**)&h->loop_dbg, (size_t)nwg * 64 * 8));
    HIP_TRY(hipMemsetAsync(h->loop_dbg, 0, (size_t)nwg * 64 * 8, s));
    h->loop_dbg_phase = phase;
    const int rc = run_loop(h, 0, x, noise, k_step, 0, s);
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(out, h->loop_dbg, (size_t)nwg * 64 * 8, hipMemcpyDeviceToHost);
    (void)hipFree(h->loop_dbg);
    h->loop_dbg = nullptr;
    if (rc != DSD_OK) return rc;
    if (e != hipSuccess) return fail(DSD_ERR_HIP, "dsd_debug_loop_timeline: %s", hipGetErrorString(e));
    *n_wg = nwg;
    return DSD_OK;
}

extern "C" int dsd_sample_ddpm(dsd_handle* h, float* x, const float* noise, int32_t k_step, void* stream) {
    DSD_TRY(check_ready(h, "dsd_sample_ddpm", true));
    if (!x) return fail(DSD_ERR_INVALID, "dsd_sample_ddpm: null argument");
    if (k_step < 1 || k_step > h->n_sched) return fail(DSD_ERR_INVALID, "dsd_sample_ddpm: k_step=%d outside 1..%d", k_step, h->n_sched);
    HIP_TRY(hipSetDevice(h->device));
    DSD_TRY(build_step_table(h, h->n_sched, (hipStream_t)stream));
    return run_loop(h, 0, x, noise, k_step, 0, (hipStream_t)stream);
}

extern "C" int dsd_p_sample(dsd_handle* h, float* x, const float* noise, int32_t t, void* stream) {
    DSD_TRY(check_ready(h, "dsd_p_sample", true));
    if (!x) return fail(DSD_ERR_INVALID, "dsd_p_sample: null argument");
    if (t < 0 || t >= h->n_sched) return fail(DSD_ERR_INVALID, "dsd_p_sample: t=%d outside the %d-step schedule", t, h->n_sched);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DSD_TRY(build_step_table(h, h->n_sched, s));
    hipLaunchKernelGGL(k_set_cell, dim3(1), dim3(1), 0, s, h->noise_cell, noise);
    if (!noise) hipLaunchKernelGGL(k_set_seed, dim3(1), dim3(1), 0, s, h->seed_cell, h->noise_seed);
    HIP_TRY(hipGetLastError());
    DSD_TRY(ensure_cp(h, false, s));
    DSD_TRY(launch_inproj(h, x, s));
    DSD_TRY(launch_stack(h, t, nullptr, s));
    HeadParams p = head_base(h);
    p.x_base = x; p.x_out = x; p.noise_off = 0; p.step_id = 0;
    p.sa = h->tab[6][t]; p.sb = h->tab[7][t]; p.c1 = h->tab[10][t]; p.c2 = h->tab[11][t];
    p.sigma = (t == 0) ? 0.f : std::exp(0.5f * h->tab[9][t]);
    return launch_head<HEAD_DDPM>(h, p, false, s);
}

// p_sample with everything the reference's signature allows (shallow_diffusion_tts.py:159-166): a step index PER UTTERANCE,
// clip_denoised on / off, and noise either per utterance or one [M][T] draw repeated over the batch (repeat_noise, noise_like :38-41).
// = DiffNet evaluation with t[B] (the per-layer kernels) + one element-wise kernel with per-utterance coefficients.
extern "C" int dsd_p_sample_ex(dsd_handle* h, float* x, const float* noise, const int32_t* t, int32_t clip_denoised, int32_t repeat_noise,
                               void* stream) {
    DSD_TRY(check_ready(h, "dsd_p_sample_ex", true));
    if (!x || !noise || !t) return fail(DSD_ERR_INVALID, "dsd_p_sample_ex: null argument");
    for (int b = 0; b < h->B; ++b)
        if (t[b] < 0 || t[b] >= h->n_sched) return fail(DSD_ERR_INVALID, "dsd_p_sample_ex: t[%d]=%d outside the %d-step schedule", b, t[b], h->n_sched);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t bmt = (size_t)h->B * h->M * h->T;
    if (!h->eps_tmp) DSD_TRY(dev_alloc(h, &h->eps_tmp, (size_t)h->cap_spec, true));
    DSD_TRY(dsd_denoise(h, x, t, h->eps_tmp, stream));
    char* pin = nullptr; int slot = 0;
    DSD_TRY(pin_acquire(h, (size_t)h->B * 5 * 4, &pin, &slot));
    float* c = reinterpret_cast<float*>(pin);
    for (int b = 0; b < h->B; ++b) {
        const int tt = t[b];
        c[5 * b + 0] = h->tab[6][tt]; c[5 * b + 1] = h->tab[7][tt]; c[5 * b + 2] = h->tab[10][tt]; c[5 * b + 3] = h->tab[11][tt];
        c[5 * b + 4] = (tt == 0) ? 0.f : std::exp(0.5f * h->tab[9][tt]);
    }
    HIP_TRY(hipMemcpyAsync(h->coef_dev, pin, (size_t)h->B * 5 * 4, hipMemcpyHostToDevice, s));
    DSD_TRY(pin_release(h, slot, s));
    const size_t per = (size_t)h->M * h->T;
    hipLaunchKernelGGL(k_psample_ex, dim3((unsigned)std::min<size_t>((bmt + 255) / 256, 8192)), dim3(256), 0, s, x, h->eps_tmp, noise, h->coef_dev,
                       per, bmt, clip_denoised ? 1 : 0, repeat_noise ? (size_t)0 : per);
    HIP_TRY(hipGetLastError());
    return DSD_OK;
}

extern "C" int dsd_sample_plms(dsd_handle* h, float* x, int32_t k_step, int32_t interval, void* stream) {
    DSD_TRY(check_ready(h, "dsd_sample_plms", true));
    if (!x) return fail(DSD_ERR_INVALID, "dsd_sample_plms: null argument");
    if (k_step < 1 || k_step > h->n_sched) return fail(DSD_ERR_INVALID, "dsd_sample_plms: k_step=%d outside 1..%d", k_step, h->n_sched);
    if (interval < 1) return fail(DSD_ERR_INVALID, "dsd_sample_plms: interval must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    DSD_TRY(build_step_table(h, h->n_sched, (hipStream_t)stream));
    return run_loop(h, 1, x, nullptr, k_step, interval, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------
// measurement hook
// ------------------------------------------------------------------------------------------------------------
extern "C" int dsd_time_layer_kernel(dsd_handle* h, int32_t layer, int32_t t, int32_t iters, float* avg_ms, void* stream) {
    DSD_TRY(check_ready(h, "dsd_time_layer_kernel", false));
    if (!avg_ms || iters < 1 || layer >= h->L || t < 0) return fail(DSD_ERR_INVALID, "dsd_time_layer_kernel: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DSD_TRY(build_step_table(h, t + 1, s));
    DSD_TRY(ensure_cp(h, false, s));
    // The launches are timed the way the sampling loop issues them: as nodes of ONE hipGraph (eager launches carry a
    // cache write-back / invalidate between kernels that graph nodes do not, ~10 % on this kernel).  layer < 0 walks
    // the non-last layers 0..L-2 in order, like one denoiser evaluation does.
    const int nl = std::max(h->L - 1, 1);
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = DSD_OK;
    for (int i = 0; i < iters && rc == DSD_OK; ++i) rc = launch_layer(h, layer >= 0 ? layer : i % nl, t, nullptr, h->cap_stream);
    const hipError_t ec = hipStreamEndCapture(h->cap_stream, &g);
    if (rc != DSD_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (ec != hipSuccess) return fail(DSD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ec));
    hipGraphExec_t ge = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) return fail(DSD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipGraphLaunch(ge, s);            // warm-up replay
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    if (e == hipSuccess) e = hipGraphLaunch(ge, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge);
    if (e != hipSuccess) return fail(DSD_ERR_HIP, "dsd_time_layer_kernel: %s", hipGetErrorString(e));
    *avg_ms = ms / (float)iters;
    return DSD_OK;
}

// Debug hook: per-wave s_memtime stamps of ONE launch of layer `layer` (start, staged, conv done, gate done, out-proj
// done, end) -> HOST out[blocks*4*8] (u64).  *n_blocks receives the grid size.
extern "C" int dsd_debug_layer_timeline(dsd_handle* h, int32_t layer, int32_t t, uint64_t* out, int32_t max_blocks, int32_t* n_blocks,
                                        void* stream) {
    DSD_TRY(check_ready(h, "dsd_debug_layer_timeline", false));
    if (!out || !n_blocks || layer < 0 || layer >= h->L || t < 0) return fail(DSD_ERR_INVALID, "dsd_debug_layer_timeline: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DSD_TRY(build_step_table(h, t + 1, s));
    DSD_TRY(ensure_cp(h, false, s));
    const int nb = layer_nb(h);
    const int blocks = h->B * ((h->ntile32 + nb - 1) / nb);
    if (blocks > max_blocks) return fail(DSD_ERR_INVALID, "dsd_debug_layer_timeline: %d blocks > buffer %d", blocks, max_blocks);
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)blocks * 32 * 8));
    // stamped launch as the LAST node of a small hipGraph (preceded by the layers that precede it in an evaluation),
    // so the stamps see the cache / dispatch conditions of the sampling loop rather than those of eager launches
    {
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = DSD_OK;
        for (int l = std::max(0, layer - 3); l < layer && rc == DSD_OK; ++l) rc = launch_layer(h, l, t, nullptr, h->cap_stream);
        if (rc == DSD_OK) rc = launch_layer(h, layer, t, nullptr, h->cap_stream, d);
        const hipError_t ec = hipStreamEndCapture(h->cap_stream, &g);
        if (rc != DSD_OK) { if (g) (void)hipGraphDestroy(g); (void)hipFree(d); return rc; }
        if (ec != hipSuccess) { (void)hipFree(d); return fail(DSD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ec)); }
        hipGraphExec_t ge = nullptr;
        const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { (void)hipFree(d); return fail(DSD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
        hipError_t e = hipGraphLaunch(ge, s);
        if (e == hipSuccess) e = hipGraphLaunch(ge, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipGraphExecDestroy(ge);
        if (e != hipSuccess) { (void)hipFree(d); return fail(DSD_ERR_HIP, "dsd_debug_layer_timeline: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipMemcpy(out, d, (size_t)blocks * 32 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    *n_blocks = blocks;
    return DSD_OK;
}

#include "fs2_abi.hpp"
#include "train_abi.hpp"
#include "voc_abi.hpp"
