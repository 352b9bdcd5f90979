/*
 * dmrecon_host.cpp -- host side of libmi_dmrecon.so: the C ABI of include/mi_dmrecon.h.
 *
 * Keeps on the host only the serial, cheap stages of mvs::DMRecon::start()
 * (reference: libs/dmrecon/dmrecon.cc): analyzeFeatures (:178-208), GlobalViewSelection
 * (global_view_selection.cc:33-101) and the feature -> seed conversion of processFeatures
 * (:243-296).  Everything per-pixel runs in the kernels of dmrecon_device.hip.
 *
 * Compiled with -ffp-contract=off so that the discrete decisions taken here
 * (frustum tests, pixel rounding, the greedy arg-max of the view selection) see the
 * same float values as the reference's unfused x86 arithmetic.
 *
 * There is no CPU fallback: every entry point that computes needs a HIP device and
 * fails with MI_DMRECON_EDEVICE otherwise.
 */
#include "../../include/mi_dmrecon.h"
#include "../../include/mi_dmrecon_debug.h"

#include <hip/hip_runtime.h>
#include <omp.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <cerrno>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <immintrin.h>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "dmrecon_device.h"
#include "pointset_device.h"
#include "gvs_device.h"
#include "dmrecon_types.h"

namespace {

/* One hardware queue per forked context / host thread: with the ROCm default of 4, streams share queues and the
 * tail launches of one context wait behind another context's bulk kernels.  Only effective if this library is
 * loaded before the HIP runtime initialises; never overrides the user's setting. */
struct HwQueueDefault { HwQueueDefault() { setenv("GPU_MAX_HW_QUEUES", "8", 0); } } g_hw_queue_default;

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(MI_DMRECON_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

/* Host threads a planning loop may use: the cores the process can actually run on -- a container may show 256 cores to
 * omp_get_num_procs() and give the process tree the CPU time of 16 (cgroup cpu.max: the GPU boxes of this project) --, at
 * most 64.  Read once. */
int host_threads_cap() {
    static const int cap = [] {
        int n = std::max(1, omp_get_num_procs());
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                 /* cgroup v2: "<quota> <period>" or "max <period>" */
            char q[32] = {0}; double per = 0.0;
            if (std::fscanf(f, "%31s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0.0)
                n = std::min(n, std::max(1, (int)std::ceil(std::atof(q) / per)));
            std::fclose(f);
        }
        return std::min(n, 64);
    }();
    return cap;
}

/* ---- small float helpers with the accumulation order of libs/math (vector.h:434-458,542-551;
 *      matrix.h:475-493): left-to-right sums starting from T(0). */
/* Waiting for the GPU.  A blocking hipStreamSynchronize / hipEventSynchronize puts the thread to sleep and the wake-up can
 * come late: measured on a 256-core host, the wait for a 3.8 ms kernel (the device view selection of a 400-view batch) took
 * 4 to 29 ms, and the same bench ran 5 % faster under rocprofv3, whose helper thread keeps the completion signals warm.
 * The round loops wait for work that is microseconds to a few milliseconds away: poll, with a pause between looks, and
 * only fall back to the blocking call when the wait gets long. */
double now_ms();
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
/* A LARGE batch (set for the calling thread by PatientWaits) waits for rounds of milliseconds: after a short spin it naps
 * 50 us between looks -- a wait ends at most one nap late (25 waits per batch of 270 ms), and the thread does not burn a
 * core per call in flight: eight ranks of a node, or four processes on one GPU, each spinning through their batches add
 * up to more CPU time than a container may have (the GPU boxes of this project: the time of 16 cores). */
thread_local bool g_patient_waits = false;
struct PatientWaits {
    bool old;
    explicit PatientWaits(bool on) : old(g_patient_waits) { g_patient_waits = on; }
    ~PatientWaits() { g_patient_waits = old; }
};
inline void wait_pause(unsigned spins) {
    if (g_patient_waits && spins >= 64u) std::this_thread::sleep_for(std::chrono::microseconds(50));   /* (64 looks: ~100 us) */
    else for (int k = 0; k < 32; ++k) cpu_relax();
}
inline hipError_t wait_event(hipEvent_t e) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        wait_pause(spins);
        if (!g_patient_waits && (spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) return hipEventSynchronize(e);
    }
}
inline hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q;
        wait_pause(spins);
        if (!g_patient_waits && (spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) return hipStreamSynchronize(s);
    }
}

struct V3 { float v[3]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
inline V3 mk(float a, float b, float c) { V3 r; r.v[0] = a; r.v[1] = b; r.v[2] = c; return r; }
inline float dot3(const float* a, const float* b) { return ((0.f + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]; }
inline V3 sub(V3 const& a, V3 const& b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline float norm3(V3 const& a) { return std::sqrt(dot3(a.v, a.v)); }
inline V3 normalized(V3 const& a) { float n = norm3(a); return mk(a[0] / n, a[1] / n, a[2] / n); }
inline V3 xform(const float* w2c, V3 const& p) {          /* Matrix4f::mult(Vec3f, 1) */
    V3 r;
    for (int i = 0; i < 3; ++i) r[i] = dot3(w2c + 4 * i, p.v) + 1.f * w2c[4 * i + 3];
    return r;
}
inline float mround(float x) { return x > 0.f ? std::floor(x + 0.5f) : std::ceil(x - 0.5f); }   /* math/functions.h:68-73 */
const float kPi = 3.141592653589793f;

struct HostLevel { int w, h; float proj[9], invproj[9]; uint32_t tex_off; };

struct HostView {
    bool valid = false;
    mi_dmrecon_camera cam;
    float cam_pos[3];
    float w2c[16];
    std::vector<HostLevel> levels;
    uint32_t* d_img = nullptr;               /* RGBA8 levels | footprint records (DevView::img / ::quad) */
    size_t n_texels = 0, quad_off = 0;       /* texels over all levels; offset (in dwords) of the records */

    V3 pos() const { return mk(cam_pos[0], cam_pos[1], cam_pos[2]); }
    /* SingleView::pointInFrustum, single_view.cc:106-119 */
    bool pointInFrustum(V3 const& wp) const {
        V3 cp = xform(w2c, wp);
        if (cp[2] <= 0.0f) return false;
        const float* P = levels[0].proj;
        float sx = dot3(P, cp.v), sy = dot3(P + 3, cp.v), sz = dot3(P + 6, cp.v);
        float x = sx / sz - 0.5f, y = sy / sz - 0.5f;
        return x >= 0 && x <= levels[0].w - 1 && y >= 0 && y <= levels[0].h - 1;
    }
    float footPrint(V3 const& p, int lvl) const { return xform(w2c, p)[2] * levels[lvl].invproj[0]; }
};

/* CameraInfo::fill_calibration / fill_inverse_calibration, libs/mve/camera.cc:124-144,179-200 */
void calibration(mi_dmrecon_camera const& c, float width, float height, float* K, float* Ki) {
    float dim_aspect = width / height;
    float image_aspect = dim_aspect * c.paspect;
    float ax, ay;
    if (image_aspect < 1.0f) { ax = c.flen * height / c.paspect; ay = c.flen * height; }
    else { ax = c.flen * width; ay = c.flen * width * c.paspect; }
    K[0] = ax; K[1] = 0.f; K[2] = width * c.ppoint[0];
    K[3] = 0.f; K[4] = ay; K[5] = height * c.ppoint[1];
    K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
    Ki[0] = 1.0f / ax; Ki[1] = 0.f; Ki[2] = -width * c.ppoint[0] / ax;
    Ki[3] = 0.f; Ki[4] = 1.0f / ay; Ki[5] = -height * c.ppoint[1] / ay;
    Ki[6] = 0.f; Ki[7] = 0.f; Ki[8] = 1.f;
}

struct Feature { float pos[3]; int ref_begin, ref_end; };

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        /* a buffer that has to grow grows to twice what is asked for (a quarter more beyond 4 GB): the calls of a scene
         * come in a few sizes (one caller's views, two or three callers merged), and hipFree + hipMalloc inside a call --
         * hipFree synchronises the device -- stalls the batch that runs next to it as well */
        size_t want = (n * sizeof(T) > ((size_t)4 << 30) ? n + n / 4 : 2 * n) + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

/* rounds of the propagation tail enqueued per read-back (even: the list buffers ping-pong per round) */
#define MI_TAIL_CHUNK 32
struct TailPoll { unsigned rw[MI_TAIL_CHUNK]; DevCounters hc; };
/* reconstruct calls in progress per device, all contexts of the process: a call that shares the GPU with several
 * others runs its small tail rounds in the one-wavefront form (BatchRun::tail_rounds) */
#define MI_MAX_DEVICES 64
std::atomic<int> g_active_calls[MI_MAX_DEVICES];
std::atomic<int> g_inject_footprint(-1);     /* test hook, see fill_job */
/* MI_DMRECON_FRONT defaults: entries per reference view (average over the batch) below which the rest of the
 * propagation goes to the front kernel, for a call alone on its GPU / next to other calls (BatchRun::tail_rounds) */
#define MI_MERGE_SMALL_CALL 48      /* reference views: below this a call waits MI_MERGE_WINDOW_US for company, ... */
#define MI_MERGE_WINDOW_US 1000
#define MI_MERGE_WINDOW_BIG_US 3000 /* ... from this size on this long: ~2 % of such a call's own time (see mi_dmrecon_reconstruct) */
#define MI_SINGLE_FOLLOW 4         /* follow-up launches of one attempt per entry in a large round (BatchRun::bulk_rounds) */
#define MI_FAST_FOLLOW 0           /* of them, the first n in the FAST kernel (MI_DMRECON_FAST_FOLLOW) */
#define MI_FOLLOW_LISTS 64         /* follow-up list counters per round: five lists x MI_FOLLOW_SEGS segments in use (BatchRun::bulk_rounds) */
#define MI_ONE_LAUNCH_MAX 100000u  /* host-visible rounds below this many entries: one launch instead of first + follow-up */
#define MI_SPEC_ROUNDS 400000u      /* throughput rounds below this many entries try an entry's candidate hypotheses at the same time */
#define MI_VIEW_HANDOVER 320u      /* a view leaves the throughput layout once a round's list of its own is shorter than this */
#define MI_TEAM_WAIT_US 20000u     /* a front team member waits this long for the others before the team gives up */
#define MI_FRONT_MIN_CAP 256       /* hand-over to k_front: entries per view, at least */
#define MI_FRONT_PER_TEAM_WG 64    /* ... and per workgroup of a view's team */
struct ActiveCall {
    int dev;
    explicit ActiveCall(int d) : dev(d >= 0 && d < MI_MAX_DEVICES ? d : -1) { if (dev >= 0) g_active_calls[dev].fetch_add(1); }
    ~ActiveCall() { if (dev >= 0) g_active_calls[dev].fetch_sub(1); }
    bool alone() const { return dev < 0 || g_active_calls[dev].load() <= 1; }
    int count() const { return dev < 0 ? 1 : g_active_calls[dev].load(); }
};

struct JobHost {          /* host-side plan of one reference view */
    int ref_view = -1;
    int status = MI_DMRECON_OK;
    int w = 0, h = 0;
    std::vector<int> global;                 /* ascending view ids */
    std::vector<DevEntry> seeds;
    std::vector<DevHyp> seed_hyp;
    size_t n_seeds = 0;
    size_t pix_off = 0;                      /* offset of this job's maps in the batch arrays */
};

}  // namespace

/* What mve::Scene + ImagePyramidCache are to the reference: the views (pyramids resident in HBM),
 * the bundle features and the lookup table.  Shared, read-only, by a context and its forks. */
/* What analyzeFeatures (dmrecon.cc:178-208) and benefitFromView (global_view_selection.cc:62-101) compute from the
 * scene alone -- not from the reference view or the settings: which view sees which feature, the feature's depth
 * in each view (-> footPrint) and the parallax between two views at a feature.  Built once per scene (like the image
 * pyramids) so that the global view selection of a reference view is table look-ups plus its greedy loop; the float
 * operations behind every entry are the ones the per-view code performs, so the selection is bit-identical. */
/* The parallax of the features two views share, for bundles too large for the dense tables below: per pair of views the
 * features both are attached to (ascending) and the angle between their directions camera -> feature (parallax(),
 * mvs_tools.h:46-56) -- what the greedy loop of the global view selection asks for once per selected view, candidate and
 * feature, for every reference view that meets the pair.  Built on first use by whichever planning thread asks first,
 * immutable afterwards, dropped with the scene's tables.  A scene whose pairs outgrow the budget goes on computing. */
struct PairList { std::vector<int> f; std::vector<float> plx; };
struct PairCache {
    static constexpr int SHARDS = 64;
    static constexpr size_t BUDGET = (size_t)1 << 30;
    std::mutex mu[SHARDS];
    std::unordered_map<uint64_t, std::shared_ptr<const PairList> > map[SHARDS];
    std::atomic<size_t> bytes{0};
};

struct SceneGeom {
    bool built = false, has_plx = false;
    std::shared_ptr<PairCache> pairs;        /* the direct form's cache (has_plx == false) */
    bool on_device = false;                  /* the tables below are in SceneStore::d_geom_* (gvs_device.hip) */
    size_t nv = 0, nf = 0;
    std::vector<uint8_t> sees;               /* [v * nf + f]: v references f and f is inside v's frustum */
    std::vector<uint8_t> refs;               /* [v * nf + f]: v references f (Feature::contains_view_id) */
    std::vector<float> zcam;                 /* [v * nf + f]: (worldToCam_v . f).z */
    std::vector<float> plx;                  /* [(v1 * nv + v2) * nf + f]: parallax in degrees where both see f, +inf elsewhere */
    /* [(v1 * nv + v2) * nf + f]: the factor benefitFromView multiplies a score by for the pair (:76-79,91-98) -- (plx / 10)^2
     * below minParallax, 1 elsewhere -- for the minParallax of the calls so far (a setting; entries are immutable once
     * built and shared by the threads that plan with them; guarded by SceneStore::mu) */
    struct PairFactors {
        std::vector<float> f;                /* [(v1 * nv + v2) * nf + f] */
        std::vector<uint8_t> plain;          /* [v1 * nv + v2]: every factor of the pair is 1 (the two views are nowhere closer than
                                              * minParallax: most pairs) -- multiplying by the row changes nothing */
    };
    std::vector<std::pair<float, std::shared_ptr<const PairFactors> > > pen;
    /* A COMPONENT's tables (SceneStore::sub): the views and features of one connected component of the bundle's view-feature
     * graph, ascending -- local index -> scene index; empty = the tables are the whole scene's (identity) */
    std::vector<int> vmap, fmap;
    size_t view(size_t local) const { return vmap.empty() ? local : (size_t)vmap[local]; }
    size_t feat(size_t local) const { return fmap.empty() ? local : (size_t)fmap[local]; }
};

/*
 * Everything a batch allocates on top of the resident scene: lists, results, state maps (~200 B per pixel of the batch:
 * gigabytes for a merged batch of hundreds of views), job records, mailboxes, pinned poll buffers, events.  It belongs
 * to the SCENE and is leased to a context for the duration of a call (ScratchLease): which of the forked contexts runs
 * the large merged batch of a step is a matter of arrival order, and a context that meets its first 200-view batch
 * inside a timed region used to pay hipFree + hipMalloc of its own buffers there -- 0.2 s, and hipFree synchronises
 * the device, so the batch running next to it stalled too (the bench's plan: 700 instead of 1 180 depth-maps/s in one
 * run out of ten).
 */
struct BatchScratch {
    DevBuf<DevJob> d_jobs;
    DevBuf<DevEntry> d_work;
    DevBuf<DevEntry> d_work2;                /* ping-pong partner of d_work in the tail rounds */
    DevBuf<DevHyp> d_hyp;
    DevBuf<DevResult> d_results;
    DevBuf<DevResult> d_results2;            /* ping-pong partner of d_results in the tail rounds */
    DevBuf<float> d_maps;                    /* depth | dz | conf | normal per batch */
    DevBuf<uint32_t> d_imaps;                /* views | upd */
    DevBuf<uint32_t> d_jobs_packed;          /* the used parts of a batch's job records as they come up, before mi_launch_unpack_jobs spreads them */
    DevBuf<uint32_t> d_xviews;               /* nrReconNeighbors > 8 only: view slots 8..15 of the sets -- per pixel | per list entry | per explicit hypothesis, two words each (DevJob::views_x) */
    DevBuf<unsigned long long> d_keys;
    DevBuf<unsigned> d_keyoff;
    DevBuf<unsigned> d_round_work;           /* [MI_MAX_ROUNDS] per round: size of the list of the views in the latency layout (host-visible
                                              * rounds), accepted entries (fused tail rounds) */
    DevBuf<unsigned> d_round_work_t;         /* [MI_MAX_ROUNDS] per round: size of the list of the views in the throughput layout */
    DevBuf<unsigned> d_round_items;          /* [MI_MAX_ROUNDS] per round: (entry, candidate) pairs of that list (speculative rounds) */
    DevBuf<unsigned> d_view;                 /* k_generate: [3][n_jobs] entries per view of the last rounds | [n_jobs] hand-over rounds */
    DevBuf<unsigned> d_front_order;          /* k_front, one workgroup per view: the view every block runs (FrontArgs::job_order) */
    DevBuf<unsigned> d_front_map;            /* k_front with teams: what every block of the grid is (FrontArgs::block_map) */
    DevBuf<unsigned> d_front;                /* k_front: [n_jobs] list offsets | [n_jobs] list sizes | [n_jobs][4] per-view statistics |
                                              * [n_jobs] pixels filled by the view's team */
    DevBuf<unsigned long long> d_front_resume;   /* k_front: [2][n_jobs] where a view goes on (FrontArgs::job_resume / job_start) */
    DevBuf<unsigned long long> d_front_mail; /* k_front teams: a mailbox per view (MI_FRONT_MAIL_WORDS) */
    DevBuf<unsigned> d_front_flags;          /* ... and MI_FRONT_TEAM_MAX pass flags per view */
    DevBuf<unsigned> d_follow;               /* 4 x work-list capacity: entries that continue with their next hypothesis (two-launch
                                              * rounds) / the (entry, candidate) items of a speculative round (at most four per entry) */
    DevBuf<unsigned> d_follow_cnt;           /* [MI_MAX_ROUNDS][MI_FOLLOW_LISTS] sizes of the follow-up lists */
    DevBuf<unsigned long long> d_fmask;      /* per wavefront unit of a large round's first launch: the patches that go on (OptArgs::follow_mask) */
    DevBuf<unsigned> d_fblk;                 /* k_follow_count's sums */
    DevBuf<DevSpec> d_spec;                  /* speculative small rounds: four attempt records per entry (BatchRun::bulk_rounds) */
    TailPoll* h_poll = nullptr;              /* pinned: read-back of two tail chunks in flight */
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    uint8_t* h_dyn = nullptr;                /* pinned: read-backs of the jobs' flags / n_filled words (JobDyn), three slots */
    size_t h_dyn_cap = 0;
    unsigned* h_done = nullptr;              /* pinned, written by k_front: views that have run to their end */
    size_t h_done_cap = 0;
    uint8_t* h_up = nullptr;                 /* pinned staging of a call's uploads: job table | list offsets | seeds | their hypotheses
                                              * (a copy from pageable memory is staged by the runtime, with waits of its own) */
    size_t h_up_cap = 0;
    uint32_t* h_sparse = nullptr;            /* pinned, written by k_emit_changed: the pixels the front changed after the maps' snapshot */
    size_t h_sparse_cap = 0;                 /* ... in words */
    unsigned* h_emit_end = nullptr;          /* pinned: [views] the list's length after each view's records (copied behind its kernel) */
    size_t h_emit_cap = 0;
    DevBuf<unsigned> d_sparse_count;         /* the list's length on the device */
    int32_t* h_gvs = nullptr;                /* pinned: the device view selection's result (a copy into pageable memory is made by the
                                              * runtime with a wait of its own, in steps of 10 ms) */
    size_t h_gvs_cap = 0;
    std::vector<int32_t> h_jobdyn;           /* staging of the flag words written to dead jobs */
    DevBuf<int32_t> d_gvs_feat, d_gvs_out;   /* scratch and result of the device view selection */
    DevBuf<float> d_gvs_base, d_gvs_benefit;
    DevBuf<GvsRef> d_gvs_refs;
    std::vector<hipEvent_t> events;
    size_t pixels() const { return d_maps.cap / 14; }       /* state maps: 14 floats per pixel */
    /* the buffers whose size goes with the pixels of a batch (imaps: 4 words per pixel, 6 with eight view slots) */
    int reserve_pixels(size_t px, size_t n_imaps) {
        return d_maps.reserve(px * 14) || d_imaps.reserve(px * n_imaps) || d_work.reserve(px) || d_work2.reserve(px)
            || d_results.reserve(px) || d_results2.reserve(px) || d_keys.reserve(px) || d_follow.reserve(4 * px + 4096);
    }
    /* room for a batch of `px` pixels: all pixel-proportional buffers together, so that a set is either large enough or
     * grows once */
    int ensure_pixels(size_t px, size_t n_imaps) {
        if (d_maps.cap >= px * 14 && d_imaps.cap >= px * n_imaps && d_work.cap >= px && d_work2.cap >= px && d_results.cap >= px
            && d_results2.cap >= px && d_keys.cap >= px && d_follow.cap >= 4 * px + 4096)
            return 0;
        return reserve_pixels(px, n_imaps);                           /* (DevBuf::reserve adds the headroom) */
    }
    bool holds_anything() const {
        return d_maps.cap || d_work.cap || d_jobs.cap || d_results.cap || d_hyp.cap || d_spec.cap || d_xviews.cap || d_gvs_out.cap || d_gvs_feat.cap
            || h_poll || h_dyn || h_gvs || h_up || h_done || h_sparse || h_emit_end || !events.empty();
    }
    void release() {
        for (size_t i = 0; i < events.size(); ++i) (void)hipEventDestroy(events[i]);
        events.clear();
        d_jobs.release(); d_work.release(); d_work2.release(); d_hyp.release(); d_results.release(); d_results2.release();
        d_follow.release(); d_follow_cnt.release(); d_fmask.release(); d_fblk.release(); d_spec.release(); d_maps.release(); d_imaps.release(); d_xviews.release(); d_jobs_packed.release(); d_keys.release(); d_keyoff.release();
        d_round_work.release(); d_round_work_t.release(); d_round_items.release(); d_view.release(); d_front.release(); d_front_resume.release();
        d_front_mail.release(); d_front_flags.release(); d_front_map.release(); d_front_order.release();
        d_gvs_feat.release(); d_gvs_out.release(); d_gvs_base.release(); d_gvs_benefit.release(); d_gvs_refs.release(); d_sparse_count.release();
        if (h_sparse) (void)hipHostFree(h_sparse);
        if (h_emit_end) (void)hipHostFree(h_emit_end);
        h_sparse = nullptr; h_sparse_cap = 0; h_emit_end = nullptr; h_emit_cap = 0;
        if (h_poll) (void)hipHostFree(h_poll);
        if (h_dyn) (void)hipHostFree(h_dyn);
        if (h_done) (void)hipHostFree(h_done);
        if (h_gvs) (void)hipHostFree(h_gvs);
        if (h_up) (void)hipHostFree(h_up);
        h_gvs = nullptr; h_gvs_cap = 0; h_up = nullptr; h_up_cap = 0;
        h_poll = nullptr; h_dyn = nullptr; h_dyn_cap = 0; h_done = nullptr; h_done_cap = 0;
        for (int k = 0; k < 2; ++k) { if (poll_ev[k]) (void)hipEventDestroy(poll_ev[k]); poll_ev[k] = nullptr; }
    }
};

/* A reconstruct call waiting to be merged with others on the same scene (see mi_dmrecon_reconstruct) */
struct MergeReq {
    const mi_dmrecon_settings* st; int32_t n; const int32_t* refs; mi_dmrecon_maps* maps; int32_t* status; mi_dmrecon_stats* stats;
    int rc = 0; std::string err; bool done = false; bool taken = false; bool served = false;
};
struct MergeQueue {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<MergeReq*> pending;          /* requests not yet part of a running batch */
    int running = 0;                         /* batches being executed */
    bool gathering = false;                  /* a leader is waiting a moment for more requests before it starts */
    int company_credit = 0;                  /* > 0: calls of this scene have met lately -- worth a short wait for more */
    /* How many calls of this scene are under way at a time, lately (the largest figure of the last eight batches, counted when
     * a batch forms: its own calls + those still pending + those inside batches that are running): a leader that has gathered
     * all of them that CAN come -- the ones inside a running batch cannot -- does not wait for its window to run out.  (Counting
     * the calls a batch served instead would lock a pattern in: four callers that once met two and two would have gone on
     * leaving in pairs.) */
    int expect = 0, running_calls = 0, recent[8] = {0, 0, 0, 0, 0, 0, 0, 0}, recent_i = 0;
};

struct SceneStore {
    int device = 0;
    MergeQueue merge;
    std::mutex pool_mu;                      /* guards scratch_pool */
    std::vector<BatchScratch> scratch_pool;  /* the scratch sets no call holds at the moment */
    SceneGeom geom;                          /* guarded by mu; dropped with views_dirty / set_features */
    /* A bundle too large for the dense tables of `geom` that falls apart into components which share no feature (several
     * scenes resident in one context; a large reconstruction in disconnected parts): tables PER COMPONENT, where a component is
     * small enough for them.  A reference view's candidates are the views that share a feature with it -- views of its own
     * component --, so its selection from the component's tables is the selection from the scene's (global_view_selection.cc:
     * 62-101: a benefit is a sum over shared features; only a benefit above zero is ever selected, :44-52).  Built with `geom`. */
    std::vector<std::unique_ptr<SceneGeom> > sub;
    std::vector<int> sub_of_view, local_of_view;     /* per view: its component's tables (-1: none -- the direct path) and its index there */
    std::mutex mu;                           /* guards the lazy upload of the DevView table */
    std::vector<HostView> views;
    std::vector<Feature> features;
    std::vector<int> feat_refs;
    /* the features every view is referenced by, ascending (the bundle's view lists inverted; built with the features):
     * a reference view's seed candidates are the features of itself and of its global views -- the union of a few of
     * these lists -- instead of a scan of every feature of the scene */
    std::vector<int> by_view_off, by_view;
    bool views_dirty = true;
    DevBuf<DevView> d_views;
    float* d_lut = nullptr;
    DevBuf<uint8_t> d_geom_sees, d_geom_valid;               /* SceneGeom for the device view selection */
    DevBuf<float> d_geom_zcam, d_geom_plx, d_geom_fpos, d_geom_inv0;
    ~SceneStore() {
        (void)hipSetDevice(device);
        for (size_t i = 0; i < scratch_pool.size(); ++i) scratch_pool[i].release();
        for (size_t i = 0; i < views.size(); ++i) if (views[i].d_img) (void)hipFree(views[i].d_img);
        d_views.release();
        d_geom_sees.release(); d_geom_valid.release(); d_geom_zcam.release(); d_geom_plx.release();
        d_geom_fpos.release(); d_geom_inv0.release();
        if (d_lut) (void)hipFree(d_lut);
    }
};

struct mi_dmrecon_ctx {
    int device = 0;
    int n_cus = 64;                          /* compute units (queried at creation) */
    double wall_clock_khz = 100000.0;        /* rate of the constant clock wall_clock64() reads (hipDeviceAttributeWallClockRate) */
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           /* copies of finished views back to the host while the front kernel still runs */
    std::shared_ptr<SceneStore> sc;
    DevCounters* d_counters = nullptr;
    DevBuf<uint8_t> d_stage;
    DevBuf<uint8_t> d_stage2;
    int stage_flip = 0;
    BatchScratch bs;                         /* leased from the scene's pool for the duration of a call (ScratchLease) */
};

namespace {

int sync_views(mi_dmrecon_ctx* c) {
    std::lock_guard<std::mutex> lock(c->sc->mu);
    if (!c->sc->views_dirty) return 0;
    std::vector<DevView> hv(c->sc->views.size());
    for (size_t i = 0; i < c->sc->views.size(); ++i) {
        HostView const& v = c->sc->views[i];
        DevView& d = hv[i];
        std::memset(&d, 0, sizeof(d));
        if (!v.valid) continue;
        std::memcpy(d.cam_pos, v.cam_pos, sizeof(d.cam_pos));
        std::memcpy(d.w2c, v.w2c, sizeof(d.w2c));
        d.n_levels = (int)v.levels.size();
        d.img = v.d_img;
        d.quad = v.d_img ? v.d_img + v.quad_off : nullptr;
        for (int l = 0; l < d.n_levels; ++l) {
            HostLevel const& L = v.levels[l];
            d.lv[l].ax = L.proj[0]; d.lv[l].ay = L.proj[4]; d.lv[l].cx = L.proj[2]; d.lv[l].cy = L.proj[5];
            d.lv[l].inv0 = L.invproj[0]; d.lv[l].w = L.w; d.lv[l].h = L.h; d.lv[l].tex_off = L.tex_off;
        }
    }
    if (c->sc->d_views.reserve(hv.size())) return fail(MI_DMRECON_EDEVICE, "hipMalloc(views) failed");
    HIP_TRY(hipMemcpyAsync(c->sc->d_views.p, hv.data(), hv.size() * sizeof(DevView), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(wait_stream(c->stream));
    c->sc->views_dirty = false;
    return 0;
}

}  // namespace

/* the launchers of the kernels compiled for a filter width (dmrecon_device.hip, one object per width) */
extern const MiDeviceApi mi_device_api_fw3, mi_device_api_fw5, mi_device_api_fw7, mi_device_api_fw9, mi_device_api_fw11;
const MiDeviceApi* mi_device_api(int filter_width) {
    switch (filter_width) {
        case 3: return &mi_device_api_fw3;
        case 5: return &mi_device_api_fw5;
        case 7: return &mi_device_api_fw7;
        case 9: return &mi_device_api_fw9;
        case 11: return &mi_device_api_fw11;
        default: return nullptr;
    }
}

namespace {

int check_settings(const mi_dmrecon_settings* st) {
    if (!st) return fail(MI_DMRECON_EINVAL, "null settings");
    if (st->scale < 0) return fail(MI_DMRECON_EINVAL, "Invalid scale factor");            /* dmrecon.cc:41-42 */
    if (!mi_device_api(st->filterWidth)) return fail(MI_DMRECON_EINVAL, "filterWidth %d unsupported (3, 5, 7, 9 or 11)", st->filterWidth);
    if (st->nrReconNeighbors < 1 || st->nrReconNeighbors > MI_DMRECON_MAX_LOCAL_VIEWS)
        return fail(MI_DMRECON_EINVAL, "nrReconNeighbors must be in 1..%d", MI_DMRECON_MAX_LOCAL_VIEWS);
    if (st->globalVSMax < 1 || st->globalVSMax > MI_DMRECON_MAX_GLOBAL_VIEWS)
        return fail(MI_DMRECON_EINVAL, "globalVSMax must be in 1..%d", MI_DMRECON_MAX_GLOBAL_VIEWS);
    return 0;
}

inline bool contains_view(mi_dmrecon_ctx const* c, Feature const& f, int id) {
    for (int j = f.ref_begin; j < f.ref_end; ++j) if (c->sc->feat_refs[j] == id) return true;
    return false;
}
inline bool in_box(V3 const& p, const float* lo, const float* hi) {          /* math::geom::point_box_overlap */
    for (int i = 0; i < 3; ++i) if (p[i] < lo[i] || p[i] > hi[i]) return false;
    return true;
}
inline float parallax(V3 const& p, HostView const& v1, HostView const& v2) {   /* mvs_tools.h:46-56 */
    V3 d1 = normalized(sub(p, v1.pos())), d2 = normalized(sub(p, v2.pos()));
    float dp = std::max(std::min(dot3(d1.v, d2.v), 1.f), -1.f);
    return std::acos(dp) * 180.f / kPi;
}

/* The dense tables of `g` over its views and features (g.nv, g.nf; g.vmap / g.fmap: a component's, else the scene's);
 * vloc: scene view index -> index in g (-1: not in it), or null for the identity.  nt: threads. */
static void build_geom_tables(SceneStore& sc, SceneGeom& g, const int* vloc, int nt) {
    const size_t nv = g.nv, nf = g.nf;
    g.sees.assign(nv * nf, 0);
    g.refs.assign(nv * nf, 0);
    g.zcam.assign(nv * nf, 0.f);
    /* unit directions camera -> feature (parallax(), mvs_tools.h:46-56), only needed while building */
    std::vector<V3> dir(nv * nf);
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
    for (long f = 0; f < (long)nf; ++f) {
        Feature const& ft = sc.features[g.feat((size_t)f)];
        const V3 p = mk(ft.pos[0], ft.pos[1], ft.pos[2]);
        for (int j = ft.ref_begin; j < ft.ref_end; ++j) {
            const int vs = sc.feat_refs[j];                                /* the scene's view ... */
            if (vs < 0 || vs >= (int)sc.views.size()) continue;
            const int v = vloc ? vloc[vs] : vs;                            /* ... and its index in these tables */
            if (v < 0 || v >= (int)nv) continue;
            g.refs[(size_t)v * nf + f] = 1;
            if (!sc.views[vs].valid) continue;
            if (!sc.views[vs].pointInFrustum(p)) continue;                 /* dmrecon.cc:190,203 */
            g.sees[(size_t)v * nf + f] = 1;
            g.zcam[(size_t)v * nf + f] = xform(sc.views[vs].w2c, p)[2];    /* SingleView::footPrint's depth */
            dir[(size_t)v * nf + f] = normalized(sub(p, sc.views[vs].pos()));
        }
    }
    /* (+inf where the two views do not both see the feature: "parallax < minParallax" is false there, which is what
     * benefitFromView's seesFeature test amounts to, global_view_selection.cc:93-98) */
    g.plx.assign(nv * nv * nf, std::numeric_limits<float>::infinity());
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) if (nt > 1)
    for (long v1 = 0; v1 < (long)nv; ++v1)
        for (size_t v2 = (size_t)v1 + 1; v2 < nv; ++v2)
            for (size_t f = 0; f < nf; ++f) {
                if (!g.sees[(size_t)v1 * nf + f] || !g.sees[v2 * nf + f]) continue;
                float dp = std::max(std::min(dot3(dir[(size_t)v1 * nf + f].v, dir[v2 * nf + f].v), 1.f), -1.f);
                const float a = std::acos(dp) * 180.f / kPi;
                g.plx[((size_t)v1 * nv + v2) * nf + f] = a;               /* dot3 is symmetric in its arguments, */
                g.plx[(v2 * nv + (size_t)v1) * nf + f] = a;               /* bit for bit                         */
            }
}

/* Builds SceneStore::geom (see SceneGeom) and, for a bundle too large for it, SceneStore::sub.  Called with the scene mutex held. */
void build_scene_geom(SceneStore& sc) {
    SceneGeom& g = sc.geom;
    const size_t nv = sc.views.size(), nf = sc.features.size();
    g.nv = nv; g.nf = nf; g.built = true; g.on_device = false;
    g.pen.clear();
    g.vmap.clear(); g.fmap.clear();
    sc.sub.clear(); sc.sub_of_view.clear(); sc.local_of_view.clear();
    /* the size guard first: a bundle too large for the parallax table (1000 views x 1M features would need 4 TB) gets
     * no tables of the whole -- the direct path (plan_global_views) needs O(features of the reference view) */
    /* (test hook MI_DMRECON_DEBUG_TABLE_LIMIT=<floats>: the bound, so that a small merged scene counts as "too large") */
    const size_t limit = [] { const char* e = std::getenv("MI_DMRECON_DEBUG_TABLE_LIMIT"); return e && std::atoll(e) > 0 ? (size_t)std::atoll(e) : ((size_t)1 << 26); }();
    g.has_plx = nv * nv * nf <= limit;                                      /* 256 MB of floats at most */
    g.pairs.reset();
    const int nt = std::max(1, std::min(host_threads_cap(), 32));
    if (g.has_plx) { build_geom_tables(sc, g, nullptr, nt); return; }
    g.pairs = std::make_shared<PairCache>();
    std::vector<uint8_t>().swap(g.sees); std::vector<float>().swap(g.zcam); std::vector<float>().swap(g.plx);
    std::vector<uint8_t>().swap(g.refs);
    /* ... but tables per connected component of the view-feature graph (SceneStore::sub): union-find over the features' view
     * lists; a component gets tables if they fit the same bound, all of them together 1 GB */
    std::vector<int> parent(nv);
    for (size_t v = 0; v < nv; ++v) parent[v] = (int)v;
    auto find = [&](int v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
    for (size_t f = 0; f < nf; ++f) {
        int first = -1;
        for (int j = sc.features[f].ref_begin; j < sc.features[f].ref_end; ++j) {
            const int v = sc.feat_refs[j];
            if (v < 0 || v >= (int)nv) continue;
            if (first < 0) first = find(v);
            else { const int r = find(v); if (r != first) parent[r] = first; }
        }
    }
    std::vector<int> comp_id(nv, -1);
    std::vector<std::vector<int> > cviews, cfeats;
    for (size_t v = 0; v < nv; ++v) {
        const int r = find((int)v);
        if (comp_id[r] < 0) { comp_id[r] = (int)cviews.size(); cviews.emplace_back(); cfeats.emplace_back(); }
        comp_id[v] = comp_id[r];
        cviews[(size_t)comp_id[v]].push_back((int)v);                      /* ascending */
    }
    if (cviews.size() < 2) return;                                          /* one component: it is the scene, too large */
    for (size_t f = 0; f < nf; ++f)
        for (int j = sc.features[f].ref_begin; j < sc.features[f].ref_end; ++j) {
            const int v = sc.feat_refs[j];
            if (v < 0 || v >= (int)nv) continue;
            cfeats[(size_t)comp_id[v]].push_back((int)f);                  /* ascending; a feature belongs to one component */
            break;
        }
    sc.sub_of_view.assign(nv, -1); sc.local_of_view.assign(nv, -1);
    size_t budget = (size_t)1 << 28;                                        /* floats of parallax tables over all components */
    std::vector<size_t> todo;
    for (size_t k = 0; k < cviews.size(); ++k) {
        const size_t cv = cviews[k].size(), cf = cfeats[k].size();
        if (cv < 2 || cf == 0 || cv * cv * cf > limit || cv * cv * cf > budget) continue;
        budget -= cv * cv * cf;
        std::unique_ptr<SceneGeom> sg(new SceneGeom());
        sg->nv = cv; sg->nf = cf; sg->built = true; sg->has_plx = true; sg->vmap = cviews[k]; sg->fmap = cfeats[k];
        for (size_t l = 0; l < cv; ++l) { sc.sub_of_view[(size_t)cviews[k][l]] = (int)sc.sub.size(); sc.local_of_view[(size_t)cviews[k][l]] = (int)l; }
        todo.push_back(sc.sub.size());
        sc.sub.push_back(std::move(sg));
    }
    /* (the components side by side, each single-threaded: many small tables) */
    const int* vloc = sc.local_of_view.data();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) if (todo.size() > 1)
    for (long t = 0; t < (long)todo.size(); ++t) build_geom_tables(sc, *sc.sub[todo[(size_t)t]], vloc, 1);
    if (std::getenv("MI_DMRECON_TRACE"))
        fprintf(stderr, "[mi_dmrecon] view selection tables: the bundle (%zu views, %zu features) in %zu parts, %zu of them with tables of their own\n",
                nv, nf, cviews.size(), sc.sub.size());
}

/* The pair factors of the scene for one minParallax (SceneGeom::pen); built on first use. */
std::shared_ptr<const SceneGeom::PairFactors> scene_pair_factors(SceneStore& sc, SceneGeom& g, float minP) {
    std::lock_guard<std::mutex> lock(sc.mu);
    for (size_t i = 0; i < g.pen.size(); ++i) if (g.pen[i].first == minP) return g.pen[i].second;
    auto tab = std::make_shared<SceneGeom::PairFactors>();
    tab->f.resize(g.plx.size());
    const float* pl = g.plx.data();
    float* out = tab->f.data();
    const long n = (long)g.plx.size();
    const int nt = std::max(1, std::min(host_threads_cap(), 16));
#pragma omp parallel for schedule(static) num_threads(nt) if (n > (1 << 20))
    for (long k = 0; k < n; ++k) {
        /* (written without a branch so that the loop vectorises: the value where plx < minP, the bits of 1.0f elsewhere) */
        const float plx = pl[k];
        const float q = plx / 10.f;
        const float v = q * q;
        uint32_t vb; std::memcpy(&vb, &v, 4);
        const uint32_t m = (plx < minP) ? 0xFFFFFFFFu : 0u;
        const uint32_t rb = (vb & m) | (0x3F800000u & ~m);
        float r; std::memcpy(&r, &rb, 4);
        out[k] = r;
    }
    tab->plain.assign(g.nv * g.nv, 1);
    const long npair = (long)(g.nv * g.nv);
#pragma omp parallel for schedule(static) num_threads(nt) if (n > (1 << 20))
    for (long pr = 0; pr < npair; ++pr) {
        const float* row = out + (size_t)pr * g.nf;
        uint8_t one = 1;
        for (size_t f = 0; f < g.nf; ++f) if (row[f] != 1.f) { one = 0; break; }
        tab->plain[(size_t)pr] = one;
    }
    if (g.pen.size() >= 2) g.pen.erase(g.pen.begin());                       /* (whoever still plans with it holds it) */
    g.pen.push_back(std::make_pair(minP, std::shared_ptr<const SceneGeom::PairFactors>(tab)));
    return g.pen.back().second;
}

/* out[t] = p[t][0] + p[t][1] + ... + p[t][n - 1], every row summed in index order from 0.f (the order is the result:
 * float addition does not associate), eight rows at a time: blocks of 8 x 8 are transposed in registers so that ONE
 * vector addition adds the next element of all eight rows.  (x86-64-v3: AVX2 is part of the build's baseline.) */
static void sum_rows8(const float* const p[8], size_t n, float out[8]) {
    __m256 acc = _mm256_setzero_ps();
    size_t f = 0;
    for (; f + 8 <= n; f += 8) {
        const __m256 r0 = _mm256_loadu_ps(p[0] + f), r1 = _mm256_loadu_ps(p[1] + f), r2 = _mm256_loadu_ps(p[2] + f), r3 = _mm256_loadu_ps(p[3] + f);
        const __m256 r4 = _mm256_loadu_ps(p[4] + f), r5 = _mm256_loadu_ps(p[5] + f), r6 = _mm256_loadu_ps(p[6] + f), r7 = _mm256_loadu_ps(p[7] + f);
        const __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1), t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
        const __m256 t4 = _mm256_unpacklo_ps(r4, r5), t5 = _mm256_unpackhi_ps(r4, r5), t6 = _mm256_unpacklo_ps(r6, r7), t7 = _mm256_unpackhi_ps(r6, r7);
        const __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE), u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
        const __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE), u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
        /* column j = element f + j of rows 0..7 */
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u0, u4, 0x20));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u1, u5, 0x20));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u2, u6, 0x20));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u3, u7, 0x20));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u0, u4, 0x31));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u1, u5, 0x31));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u2, u6, 0x31));
        acc = _mm256_add_ps(acc, _mm256_permute2f128_ps(u3, u7, 0x31));
    }
    _mm256_storeu_ps(out, acc);
    for (; f < n; ++f) for (int t = 0; t < 8; ++t) out[t] += p[t][f];
}

/* plan_global_views from the scene tables: the same selection, without a single acos, projection or division per
 * candidate and round.  DENSE over the scene's features: a candidate's scores are an array of nf floats, 0 where the
 * feature is not attached to the reference view or not seen by the candidate -- adding +0 to a sum of non-negative
 * floats changes nothing, so the benefit is the reference's sum over the candidate's features in their order (:66-99)
 * bit for bit, and every loop below is a contiguous one.  The arrays live with the calling thread (GvsScratch). */
struct GvsScratch { std::vector<float> base, prod; std::vector<uint8_t> attached; };
/* g: the scene's tables, or those of the reference view's component (SceneStore::sub: local indices, mapped by g.view / g.feat);
 * ref_scene: the reference view's index in the scene. */
int plan_global_views_tables(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, SceneGeom& g, int ref_scene, std::vector<int>& global) {
    static thread_local GvsScratch W;
    const size_t nv = g.nv, nf = g.nf;
    const int ref = g.vmap.empty() ? ref_scene : c->sc->local_of_view[(size_t)ref_scene];
    HostView const& R = c->sc->views[ref_scene];
    const float minP = st->minParallax;
    const std::shared_ptr<const SceneGeom::PairFactors> pen_tab = scene_pair_factors(*c->sc, g, minP);
    const float* P = pen_tab->f.data();
    const uint8_t* plain = pen_tab->plain.data();
    const bool no_box = st->aabbMin[0] == -std::numeric_limits<float>::max() && st->aabbMax[0] == std::numeric_limits<float>::max()
                     && st->aabbMin[1] == -std::numeric_limits<float>::max() && st->aabbMax[1] == std::numeric_limits<float>::max()
                     && st->aabbMin[2] == -std::numeric_limits<float>::max() && st->aabbMax[2] == std::numeric_limits<float>::max();
    /* features attached to the reference view (dmrecon.cc:185-196) */
    W.attached.assign(nf, 0);
    uint8_t* att = W.attached.data();
    const uint8_t* sees_ref = &g.sees[(size_t)ref * nf];
    for (size_t f = 0; f < nf; ++f) {
        if (!sees_ref[f]) continue;
        if (!no_box) {
            Feature const& ft = c->sc->features[g.feat(f)];
            if (!in_box(mk(ft.pos[0], ft.pos[1], ft.pos[2]), st->aabbMin, st->aabbMax)) continue;
        }
        att[f] = 1;
    }
    std::vector<char> available(nv, 1);                                     /* global_view_selection.cc:23-30 */
    available[ref] = 0;
    for (size_t i = 0; i < nv; ++i) if (!c->sc->views[g.view(i)].valid) available[i] = 0;
    /* the part of benefitFromView's score that does not depend on the selected set (:76-89); 0 where the candidate
     * does not see an attached feature (dmrecon.cc:198-206) */
    W.base.resize(nv * nf); W.prod.resize(nv * nf);
    const float inv_m = R.levels[st->scale].invproj[0];
    const float* z_ref = &g.zcam[(size_t)ref * nf];
    for (size_t i = 0; i < nv; ++i) {
        if (!available[i]) continue;
        const float inv_n = c->sc->views[g.view(i)].levels[0].invproj[0];
        const float* p_ri = P + ((size_t)ref * nv + i) * nf;
        const float* z_i = &g.zcam[i * nf];
        const uint8_t* sv = &g.sees[i * nf];
        float* b = W.base.data() + i * nf;
        for (size_t f = 0; f < nf; ++f) {
            if (!(att[f] & sv[f])) { b[f] = 0.f; continue; }
            const float score = p_ri[f];                                    /* 1.f, times (plx / 10)^2 below minParallax */
            const float mfp = z_ref[f] * inv_m;
            const float nfp = z_i[f] * inv_n;
            float ratio = mfp / nfp;
            if (ratio > 2.) ratio = 2. / ratio;
            else if (ratio > 1.) ratio = 1.;
            b[f] = score * ratio;
        }
    }
    std::vector<int> selected;          /* kept sorted ascending = std::set order */
    std::vector<int> cand;
    std::vector<float> benefit(nv, 0.f);
    cand.reserve(nv);
    bool foundOne = true;
    while (foundOne && selected.size() < (size_t)st->globalVSMax) {
        cand.clear();
        for (size_t i = 0; i < nv; ++i) if (available[i]) cand.push_back((int)i);
        /* score[f] = base[f] * factor(sel_0)[f] * factor(sel_1)[f] ... in ascending view order (benefitFromView iterates
         * the std::set; a selected view that does not see the feature contributes x 1, :93), one selected view at a time
         * over all features: the same products in the same order per feature as the reference */
        const size_t ns = selected.size();
        for (size_t ci = 0; ci < cand.size(); ++ci) {
            const size_t i = (size_t)cand[ci];
            float* sc = W.prod.data() + i * nf;
            const float* b = W.base.data() + i * nf;
            const float* src = b;                                           /* (x 1 is exact: rows of ones are skipped) */
            for (size_t q = 0; q < ns; ++q) {
                const size_t pr = (size_t)selected[q] * nv + i;
                if (plain[pr]) continue;
                const float* pq = P + pr * nf;
                for (size_t f = 0; f < nf; ++f) sc[f] = src[f] * pq[f];
                src = sc;
            }
            if (src == b) std::memcpy(sc, b, nf * sizeof(float));
        }
        /* A candidate's benefit is the sum of its scores in feature order -- a chain of dependent additions, four cycles
         * each; the sums of EIGHT candidates run side by side (independent chains, each in its own order: the same floats). */
        for (size_t g0 = 0; g0 < cand.size(); g0 += 8) {
            const size_t m = std::min<size_t>(8, cand.size() - g0);
            const float* p[8];
            for (size_t t = 0; t < 8; ++t) p[t] = W.prod.data() + (size_t)cand[g0 + (t < m ? t : 0)] * nf;   /* (a short last group repeats its first candidate) */
            float b[8];
            sum_rows8(p, nf, b);
            for (size_t t = 0; t < m; ++t) benefit[(size_t)cand[g0 + t]] = b[t];
        }
        float maxBenefit = 0.f; size_t maxView = 0; foundOne = false;
        for (size_t i = 0; i < nv; ++i) {                                   /* ascending, strictly greater: the reference's tie-break */
            if (!available[i]) continue;
            if (benefit[i] > maxBenefit) { maxBenefit = benefit[i]; maxView = i; foundOne = true; }
        }
        if (foundOne) {
            selected.insert(std::upper_bound(selected.begin(), selected.end(), (int)maxView), (int)maxView);
            available[maxView] = 0;
        }
    }
    /* (a component's views ascend with the scene's: the order of the selection and its tie-breaks are the scene's) */
    global.resize(selected.size());
    for (size_t k = 0; k < selected.size(); ++k) global[k] = (int)g.view((size_t)selected[k]);
    return 0;
}

/* The argument checks of plan_global_views, shared with the device path */
int check_ref_view(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int ref) {
    const size_t nv = c->sc->views.size();
    if (ref < 0 || (size_t)ref >= nv) return fail(MI_DMRECON_EINVAL, "Master view index out of bounds");
    HostView const& R = c->sc->views[ref];
    /* (device < 0: the host-only planning hook, mi_dmrecon_debug_plan_views_host -- cameras and features, no images at all) */
    if (!R.valid || (c->device >= 0 && !R.d_img)) return fail(MI_DMRECON_EINVAL, "Invalid master view");
    if ((size_t)st->scale >= R.levels.size()) return fail(MI_DMRECON_EINVAL, "scale %d beyond pyramid of view %d", st->scale, ref);
    return 0;
}

/* MI_DMRECON_GVS_DEVICE: 0 = host, 1 = device whenever possible, unset = device when the call is large enough for the
 * launch to pay.  The host loop costs reference views x views^2 x features / threads (a greedy round per selected view, a
 * product over the selected set per candidate and feature), the kernel a launch plus much less per unit: measured, 100
 * reference views of a 100-view scene with 2000 features (2e9) 25 ms on the device against 45 ms of host threads; 20 of a
 * 20-view scene (1.6e7) 2.0 ms against 1.1 ms; and round 4, 400 reference views of the 20-view scene (3.2e8): the kernel is
 * the FIRST thing a call puts on a GPU that has just spent tens of milliseconds in the latency-bound end of the previous
 * batch, and its serial sums run at whatever clock the GPU has fallen to -- 3.8 ms under rocprofv3 (clocks held), 4 to 29 ms
 * in steps of ~10 ms without (profiles/r4_big_batch.txt) -- against a steady 5.1 ms of host threads.  Host threads also
 * cost the GPU nothing when several calls overlap. */
#define MI_GVS_DEVICE_MIN_WORK 1000000000.0
bool gvs_device_wanted(mi_dmrecon_ctx* c, int n_refs) {
    const char* e = std::getenv("MI_DMRECON_GVS_DEVICE");                /* read per call: tests switch it */
    const int mode = e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    if (mode >= 0) return mode != 0;
    const double nv = (double)c->sc->views.size();
    return (double)n_refs * nv * nv * (double)c->sc->features.size() >= MI_GVS_DEVICE_MIN_WORK;
}

/* The global view selection of n reference views in one launch of gvs_device.hip (one workgroup each) from the scene
 * tables, uploaded once per scene.  rc[i] != 0 on entry: view i is skipped.  Returns 1 when the device path does not
 * apply (tables too large for the scene, more views than the kernel's LDS flags) -- the caller falls back to the host
 * loop, which is the same selection -- 0 when `global[i]` / rc[i] are filled, < 0 on a device error. */
int plan_global_views_device(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int n, const int32_t* refs,
                             std::vector<std::vector<int> >& global, std::vector<int>& rc, std::vector<std::string>& err) {
    SceneStore& sc = *c->sc;
    {
        std::lock_guard<std::mutex> lock(sc.mu);
        if (!sc.geom.built) build_scene_geom(sc);
        SceneGeom& g = sc.geom;
        if (!g.has_plx || g.nv > 1024 || g.nf == 0) return 1;
        if (!g.on_device) {
            const size_t nv = g.nv, nf = g.nf;
            std::vector<float> fpos(3 * nf), inv0(nv, 0.f);
            std::vector<uint8_t> valid(nv, 0);
            for (size_t f = 0; f < nf; ++f) for (int k = 0; k < 3; ++k) fpos[3 * f + k] = sc.features[f].pos[k];
            for (size_t v = 0; v < nv; ++v) if (sc.views[v].valid) { valid[v] = 1; inv0[v] = sc.views[v].levels[0].invproj[0]; }
            if (sc.d_geom_sees.reserve(nv * nf) || sc.d_geom_zcam.reserve(nv * nf) || sc.d_geom_plx.reserve(nv * nv * nf)
                || sc.d_geom_fpos.reserve(3 * nf) || sc.d_geom_inv0.reserve(nv) || sc.d_geom_valid.reserve(nv))
                return fail(MI_DMRECON_EDEVICE, "hipMalloc(view selection tables) failed");
            HIP_TRY(hipMemcpyAsync(sc.d_geom_sees.p, g.sees.data(), nv * nf, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(sc.d_geom_zcam.p, g.zcam.data(), nv * nf * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(sc.d_geom_plx.p, g.plx.data(), nv * nv * nf * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(sc.d_geom_fpos.p, fpos.data(), 3 * nf * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(sc.d_geom_inv0.p, inv0.data(), nv * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(sc.d_geom_valid.p, valid.data(), nv, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(wait_stream(c->stream));
            g.on_device = true;
        }
    }
    const size_t nv = sc.geom.nv, nf = sc.geom.nf;
    std::vector<GvsRef> hr; std::vector<int> slot(n, -1);
    for (int i = 0; i < n; ++i) {
        if (rc[i]) continue;
        rc[i] = check_ref_view(c, st, refs[i]);
        if (rc[i]) { err[i] = g_err; continue; }
        GvsRef r; r.ref = refs[i]; r.inv_m = sc.views[refs[i]].levels[st->scale].invproj[0];
        slot[i] = (int)hr.size(); hr.push_back(r);
    }
    const size_t m = hr.size();
    if (m == 0) return 0;
    if (c->bs.d_gvs_refs.reserve(m) || c->bs.d_gvs_feat.reserve(m * nf) || c->bs.d_gvs_base.reserve(m * nv * nf)
        || c->bs.d_gvs_benefit.reserve(m * nv) || c->bs.d_gvs_out.reserve(m * (MI_GVS_MAX_OUT + 1)))
        return fail(MI_DMRECON_EDEVICE, "hipMalloc(view selection scratch) failed");
    GvsArgs a;
    a.sc.nv = (int)nv; a.sc.nf = (int)nf;
    a.sc.sees = sc.d_geom_sees.p; a.sc.zcam = sc.d_geom_zcam.p; a.sc.plx = sc.d_geom_plx.p; a.sc.fpos = sc.d_geom_fpos.p;
    a.sc.inv0 = sc.d_geom_inv0.p; a.sc.valid = sc.d_geom_valid.p;
    a.refs = c->bs.d_gvs_refs.p; a.minParallax = st->minParallax; a.globalVSMax = st->globalVSMax;
    a.use_box = 0;
    for (int k = 0; k < 3; ++k) {
        a.aabb_min[k] = st->aabbMin[k]; a.aabb_max[k] = st->aabbMax[k];
        if (st->aabbMin[k] != -std::numeric_limits<float>::max() || st->aabbMax[k] != std::numeric_limits<float>::max()) a.use_box = 1;
    }
    a.feat = c->bs.d_gvs_feat.p; a.base = c->bs.d_gvs_base.p; a.benefit = c->bs.d_gvs_benefit.p;
    a.out_ids = c->bs.d_gvs_out.p; a.out_n = c->bs.d_gvs_out.p + m * MI_GVS_MAX_OUT;
    const bool tr = std::getenv("MI_DMRECON_TRACE") != nullptr;
    const double tg0 = now_ms();
    HIP_TRY(hipMemcpyAsync(c->bs.d_gvs_refs.p, hr.data(), m * sizeof(GvsRef), hipMemcpyHostToDevice, c->stream));
    const double tg1 = now_ms();
    mi_gvs_launch(c->stream, a, (int)m);
    HIP_TRY(hipGetLastError());
    const double tg2 = now_ms();
    const size_t n_out = m * (MI_GVS_MAX_OUT + 1);
    if (c->bs.h_gvs_cap < n_out) {
        if (c->bs.h_gvs) (void)hipHostFree(c->bs.h_gvs);
        c->bs.h_gvs = nullptr; c->bs.h_gvs_cap = 0;
        if (hipHostMalloc((void**)&c->bs.h_gvs, 2 * n_out * sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "hipHostMalloc(view selection result) failed");
        c->bs.h_gvs_cap = 2 * n_out;
    }
    const int32_t* out = c->bs.h_gvs;
    HIP_TRY(hipMemcpyAsync(c->bs.h_gvs, c->bs.d_gvs_out.p, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    const double tg3 = now_ms();
    HIP_TRY(wait_stream(c->stream));
    if (tr) fprintf(stderr, "[mi_dmrecon] device view selection of %zu views: upload call %.3f ms, launch %.3f, read-back call %.3f, wait %.3f\n",
                    m, tg1 - tg0, tg2 - tg1, tg3 - tg2, now_ms() - tg3);
    for (int i = 0; i < n; ++i) {
        if (slot[i] < 0) continue;
        const int k = out[m * MI_GVS_MAX_OUT + slot[i]];
        global[i].assign(out + (size_t)slot[i] * MI_GVS_MAX_OUT, out + (size_t)slot[i] * MI_GVS_MAX_OUT + k);
    }
    return 0;
}

/* analyzeFeatures + GlobalViewSelection::performVS for one reference view; fills `global`.
 * Same arithmetic, operation order and tie-breaking as the reference (dmrecon.cc:178-208,
 * global_view_selection.cc:33-101), so the greedy arg-max sees the same floats.  What differs is
 * bookkeeping only: SingleView::seesFeature's linear scan (single_view.h:166-173) is a bitmap, the
 * unit directions feature->camera are computed once, and the pairwise parallax penalty of a newly
 * selected view is computed once instead of once per greedy round (multiplying by a cached factor,
 * or by 1.0f where the reference skips, gives bit-identical products). */
/* the shared features of views v1, v2 and their parallaxes (PairCache): from the cache, or built now */
std::shared_ptr<const PairList> pair_list(SceneStore& sc, PairCache* pc, int v1, int v2) {
    if (v1 > v2) std::swap(v1, v2);
    const uint64_t key = ((uint64_t)(uint32_t)v1 << 32) | (uint32_t)v2;
    const int shard = (int)((key * 0x9E3779B97F4A7C15ull) >> 58);
    if (pc) {
        std::lock_guard<std::mutex> lock(pc->mu[shard]);
        auto it = pc->map[shard].find(key);
        if (it != pc->map[shard].end()) return it->second;
    }
    auto L = std::make_shared<PairList>();
    const std::vector<int>& off = sc.by_view_off;
    if ((size_t)v2 + 1 < off.size() && v1 >= 0) {
        const V3 c1 = sc.views[v1].pos(), c2 = sc.views[v2].pos();
        int a = off[v1], b = off[v2];
        const int ea = off[(size_t)v1 + 1], eb = off[(size_t)v2 + 1];
        while (a < ea && b < eb) {
            const int fa = sc.by_view[(size_t)a], fb = sc.by_view[(size_t)b];
            if (fa < fb) ++a;
            else if (fb < fa) ++b;
            else {
                if (L->f.empty() || L->f.back() != fa) {
                    Feature const& f = sc.features[(size_t)fa];
                    const V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
                    const V3 d1 = normalized(sub(p, c1)), d2 = normalized(sub(p, c2));
                    const float dp = std::max(std::min(dot3(d1.v, d2.v), 1.f), -1.f);     /* (symmetric in its arguments, bit for bit) */
                    L->f.push_back(fa); L->plx.push_back(std::acos(dp) * 180.f / kPi);
                }
                ++a; ++b;
            }
        }
    }
    if (pc) {
        const size_t sz = L->f.size() * (sizeof(int) + sizeof(float)) + 64;
        if (pc->bytes.load(std::memory_order_relaxed) + sz <= PairCache::BUDGET) {
            std::lock_guard<std::mutex> lock(pc->mu[shard]);
            auto ins = pc->map[shard].emplace(key, L);
            if (ins.second) pc->bytes.fetch_add(sz, std::memory_order_relaxed);
            return ins.first->second;
        }
    }
    return L;
}

int plan_global_views(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int ref, std::vector<int>& global) {
    const size_t nv = c->sc->views.size();
    if (int r = check_ref_view(c, st, ref)) return r;
    HostView const& R = c->sc->views[ref];
    {
        /* the scene tables (built on first use after the scene changed; MI_DMRECON_GVS_TABLES=0: always the direct path) */
        const bool use_tables = true;
        if (use_tables) {
            {
                std::lock_guard<std::mutex> lock(c->sc->mu);
                if (!c->sc->geom.built) build_scene_geom(*c->sc);
            }
            if (c->sc->geom.has_plx) return plan_global_views_tables(c, st, c->sc->geom, ref, global);
            /* (a bundle in several parts: the tables of the reference view's component) */
            if ((size_t)ref < c->sc->sub_of_view.size() && c->sc->sub_of_view[(size_t)ref] >= 0)
                return plan_global_views_tables(c, st, *c->sc->sub[(size_t)c->sc->sub_of_view[(size_t)ref]], ref, global);
        }
    }
    /* The direct form (a bundle too large for the scene tables: hundreds of views).  Everything below is kept to the views
     * that can be selected at all -- a candidate's benefit is a sum over the features it shares with the reference view
     * (global_view_selection.cc:62-101), and only a benefit above zero is ever selected (:44-52) -- so the cost of a
     * reference view goes with the features it has and the views THEY are attached to, not with the size of the bundle
     * (no per-view arrays over all views, no scan of every feature's view list), and the parallaxes of the greedy loop -- one
     * per selected view, candidate and feature -- come from a cache per pair of views (PairCache: the dense scene tables'
     * counterpart for bundles too large to have them).
     * Same sums in the same order: the views are walked in ascending order, ties go to the lower id as in the reference. */
    /* features attached to the reference view (dmrecon.cc:185-196), local index = position in `feat`: the reference view's
     * own list of the inverted bundle (ascending, as the reference walks the bundle), or a scan of the bundle */
    std::vector<int> feat;
    auto consider = [&](size_t i) {
        Feature const& f = c->sc->features[i];
        V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
        if (!R.pointInFrustum(p)) return;
        if (!in_box(p, st->aabbMin, st->aabbMax)) return;
        feat.push_back((int)i);
    };
    if ((size_t)ref + 1 < c->sc->by_view_off.size()) {
        int last = -1;
        for (int k = c->sc->by_view_off[ref]; k < c->sc->by_view_off[(size_t)ref + 1]; ++k) {
            const int i = c->sc->by_view[(size_t)k];
            if (i != last) consider((size_t)i);               /* (a feature that lists the view twice is still one feature) */
            last = i;
        }
    } else {
        for (size_t i = 0; i < c->sc->features.size(); ++i)
            if (contains_view(c, c->sc->features[i], ref)) consider(i);
    }
    const size_t nf = feat.size();
    /* the views that see one of those features (dmrecon.cc:198-206), ascending; slot[v] = position in `act`, or -1 */
    std::vector<int> slot(nv, -1), act;
    for (size_t l = 0; l < nf; ++l) {
        Feature const& f = c->sc->features[feat[l]];
        for (int j = f.ref_begin; j < f.ref_end; ++j) {
            const int id = c->sc->feat_refs[j];
            if (id < 0 || id >= (int)nv || !c->sc->views[id].valid || slot[id] >= 0) continue;
            slot[id] = 0; act.push_back(id);
        }
    }
    std::sort(act.begin(), act.end());
    for (size_t a = 0; a < act.size(); ++a) slot[act[a]] = (int)a;
    const size_t na = act.size();
    std::vector<std::vector<int> > featInd(na);             /* per such view: local feature indices, ascending */
    std::vector<std::vector<uint8_t> > sees(na);
    for (size_t a = 0; a < na; ++a) sees[a].assign(nf, 0);
    for (size_t l = 0; l < nf; ++l) {
        Feature const& f = c->sc->features[feat[l]];
        V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
        for (int j = f.ref_begin; j < f.ref_end; ++j) {
            int id = c->sc->feat_refs[j];
            if (id < 0 || id >= (int)nv || !c->sc->views[id].valid) continue;
            const size_t a = (size_t)slot[id];
            if (c->sc->views[id].pointInFrustum(p)) { featInd[a].push_back((int)l); sees[a][l] = 1; }
        }
    }
    /* The parallax of a feature between two views -- all the greedy loop wants from it is `plx < minParallax` and, below it,
     * (plx / 10)^2 (:76-79, :91-98) -- comes from the scene's pair cache: the features two views share with their angles,
     * computed once per pair and scene instead of once per selected view, candidate, feature AND reference view (the arc
     * cosines were 5.5 of the 7 ms a reference view of a 400-view bundle took).  Both lists ascend: one walk per pair. */
    /* (held for the whole selection: a rebuild of the scene's tables from another context of the scene -- set_view, set_features --
     * drops the scene's pointer, not the cache this planner is walking) */
    std::shared_ptr<PairCache> pc_hold;
    {
        std::lock_guard<std::mutex> lock(c->sc->mu);
        if (!c->sc->geom.pairs) c->sc->geom.pairs = std::make_shared<PairCache>();
        pc_hold = c->sc->geom.pairs;
    }
    PairCache* const pc = pc_hold.get();
    auto factor = [&](float plx) -> float { return plx < st->minParallax ? (plx / 10.f) * (plx / 10.f) : 1.f; };
    /* the parallax of feature `gid` in the pair's list, the cursor moving on (a feature both views are attached to is in it) */
    auto lookup = [&](PairList const& PL, size_t& cur, int gid, int v1, int v2) -> float {
        while (cur < PL.f.size() && PL.f[cur] < gid) ++cur;
        if (cur < PL.f.size() && PL.f[cur] == gid) return PL.plx[cur];
        Feature const& f = c->sc->features[(size_t)gid];                   /* (not reached: computed as the list would have) */
        const V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
        const V3 d1 = normalized(sub(p, c->sc->views[std::min(v1, v2)].pos())), d2 = normalized(sub(p, c->sc->views[std::max(v1, v2)].pos()));
        return std::acos(std::max(std::min(dot3(d1.v, d2.v), 1.f), -1.f)) * 180.f / kPi;
    };
    /* the part of benefitFromView's score that does not depend on the selected set (:76-89) */
    std::vector<std::vector<float> > base(na);
    std::vector<char> available(na, 1);                                     /* global_view_selection.cc:23-30 */
    for (size_t a = 0; a < na; ++a) {
        const int i = act[a];
        if (i == ref || featInd[a].empty()) { available[a] = 0; continue; }   /* (no shared feature: benefit 0, never selected) */
        base[a].resize(featInd[a].size());
        const std::shared_ptr<const PairList> PL = pair_list(*c->sc, pc, ref, i);
        size_t cur = 0;
        for (size_t k = 0; k < featInd[a].size(); ++k) {
            const size_t l = featInd[a][k];
            Feature const& f = c->sc->features[feat[l]];
            V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
            float score = 1.f;
            score *= factor(lookup(*PL, cur, feat[l], ref, i));
            float mfp = R.footPrint(p, st->scale);
            float nfp = c->sc->views[i].footPrint(p, 0);
            float ratio = mfp / nfp;
            if (ratio > 2.) ratio = 2. / ratio;
            else if (ratio > 1.) ratio = 1.;
            score *= ratio;
            base[a][k] = score;
        }
    }
    std::vector<int> selected;          /* view ids, kept sorted ascending = std::set order */
    std::vector<size_t> selected_a;     /* ... and their positions in `act`, in the same order */
    /* pen[c][i][k]: factor view c (once selected) contributes to feature k of candidate i (:91-98) */
    std::vector<std::vector<std::vector<float> > > pen(na);
    std::vector<std::vector<char> > plain(na);   /* plain[c][i]: every factor of the pair is 1 -- multiplying by them changes nothing */
    std::vector<float> scr;
    bool foundOne = true;
    while (foundOne && selected.size() < (size_t)st->globalVSMax) {
        float maxBenefit = 0.f; size_t maxA = 0; foundOne = false;
        for (size_t a = 0; a < na; ++a) {
            if (!available[a]) continue;
            float benefit = 0;
            const size_t nk = featInd[a].size();
            /* a feature's score = its base x the factors of the selected views in ascending view order (:91-98): view by view
             * over all features (contiguous, and the same products in the same order), then the sum in feature order */
            scr.assign(base[a].begin(), base[a].end());
            for (size_t s = 0; s < selected_a.size(); ++s) {
                if (plain[selected_a[s]][a]) continue;
                const float* pv = pen[selected_a[s]][a].data();
                for (size_t k = 0; k < nk; ++k) scr[k] *= pv[k];
            }
            for (size_t k = 0; k < nk; ++k) benefit += scr[k];
            if (benefit > maxBenefit) { maxBenefit = benefit; maxA = a; foundOne = true; }
        }
        if (foundOne) {
            const size_t at = (size_t)(std::upper_bound(selected.begin(), selected.end(), act[maxA]) - selected.begin());
            selected.insert(selected.begin() + (std::ptrdiff_t)at, act[maxA]);
            selected_a.insert(selected_a.begin() + (std::ptrdiff_t)at, maxA);
            available[maxA] = 0;
            if (selected.size() < (size_t)st->globalVSMax) {
                pen[maxA].resize(na);
                plain[maxA].assign(na, 1);
                for (size_t a = 0; a < na; ++a) {
                    if (!available[a]) continue;
                    std::vector<float>& pv = pen[maxA][a];
                    pv.assign(featInd[a].size(), 1.f);
                    bool ones = true;
                    const std::shared_ptr<const PairList> PL = pair_list(*c->sc, pc, act[maxA], act[a]);
                    size_t cur = 0;
                    for (size_t k = 0; k < featInd[a].size(); ++k) {
                        const size_t l = featInd[a][k];
                        if (!sees[maxA][l]) continue;
                        pv[k] = factor(lookup(*PL, cur, feat[l], act[maxA], act[a]));
                        if (pv[k] != 1.f) ones = false;
                    }
                    plain[maxA][a] = ones ? 1 : 0;
                }
            }
        }
    }
    global = selected;
    return 0;
}

/* The host half of DMRecon::processFeatures (dmrecon.cc:258-296): feature -> (pixel, initDepth) */
/* by_view / by_view_off of a scene's features (SceneStore) */
void build_features_by_view(SceneStore& sc) {
    int max_id = -1;
    for (int v : sc.feat_refs) max_id = std::max(max_id, v);
    sc.by_view_off.assign((size_t)(max_id + 2), 0);
    for (int v : sc.feat_refs) if (v >= 0) ++sc.by_view_off[(size_t)v + 1];
    for (size_t v = 1; v < sc.by_view_off.size(); ++v) sc.by_view_off[v] += sc.by_view_off[v - 1];
    sc.by_view.assign(sc.feat_refs.size(), 0);
    std::vector<int> fill(sc.by_view_off.begin(), sc.by_view_off.end());
    for (size_t f = 0; f < sc.features.size(); ++f)
        for (int j = sc.features[f].ref_begin; j < sc.features[f].ref_end; ++j) {
            const int v = sc.feat_refs[j];
            if (v >= 0) sc.by_view[(size_t)fill[v]++] = (int)f;                 /* (features in ascending order per view) */
        }
}

void plan_seeds(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, JobHost& job, int job_index) {
    HostView const& R = c->sc->views[job.ref_view];
    HostLevel const& L = R.levels[st->scale];
    /* (the scene tables, where they exist, answer "does view v reference feature i" with one byte: the scan of the
     * feature's view list is the reference's own Feature::contains_view_id) */
    SceneGeom const& G = c->sc->geom;
    const bool tab = G.built && G.has_plx && G.refs.size() == G.nv * G.nf && G.nf == c->sc->features.size();
    const size_t nf = c->sc->features.size();
    /* without the tables (a scene too large for them): the candidates are the features the reference view or one of its
     * global views is referenced by (dmrecon.cc:262-270: useFeature), in the order of the bundle -- the union of their
     * by-view lists instead of a scan of every feature's view list for every one of those views */
    /* (a bundle in several parts: the byte tables of the reference view's component, its features in the order of the bundle) */
    const SceneGeom* SG = nullptr;
    if (!tab && G.built && (size_t)job.ref_view < c->sc->sub_of_view.size() && c->sc->sub_of_view[(size_t)job.ref_view] >= 0)
        SG = c->sc->sub[(size_t)c->sc->sub_of_view[(size_t)job.ref_view]].get();
    if (SG) {
        const size_t cf = SG->nf;
        const int* loc = c->sc->local_of_view.data();
        const int subk = c->sc->sub_of_view[(size_t)job.ref_view];
        std::vector<const uint8_t*> rows;
        rows.push_back(&SG->refs[(size_t)loc[job.ref_view] * cf]);
        for (size_t g = 0; g < job.global.size(); ++g)
            if (c->sc->sub_of_view[(size_t)job.global[g]] == subk) rows.push_back(&SG->refs[(size_t)loc[job.global[g]] * cf]);
        for (size_t it = 0; it < cf; ++it) {
            bool use = false;
            for (size_t r = 0; !use && r < rows.size(); ++r) use = rows[r][it] != 0;
            if (!use) continue;
            Feature const& f = c->sc->features[(size_t)SG->fmap[it]];
            V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
            if (!R.pointInFrustum(p)) continue;
            if (!in_box(p, st->aabbMin, st->aabbMax)) continue;
            ++job.n_seeds;
            V3 cp = xform(R.w2c, p);                                            /* worldToScreenScaled */
            float sx = dot3(L.proj, cp.v), sy = dot3(L.proj + 3, cp.v), sz = dot3(L.proj + 6, cp.v);
            int const x = (int)mround(sx / sz - 0.5f);
            int const y = (int)mround(sy / sz - 0.5f);
            if (x < 0 || y < 0 || x >= L.w || y >= L.h) continue;              /* the sampler's border test fails anyway */
            DevEntry e; e.job = job_index; e.xy = x | (y << 16);
            DevHyp h; h.depth = norm3(sub(p, R.pos())); h.dzI = 0.f; h.dzJ = 0.f; h.views = 0xFFFFFFFFu; h.views_hi = 0xFFFFFFFFu;
            job.seeds.push_back(e);
            job.seed_hyp.push_back(h);
        }
        return;
    }
    std::vector<int> cand;
    const bool by_view = !tab && c->sc->by_view_off.size() > 1;
    if (by_view) {
        auto add = [&](int v) {
            if (v < 0 || (size_t)v + 1 >= c->sc->by_view_off.size()) return;
            cand.insert(cand.end(), c->sc->by_view.begin() + c->sc->by_view_off[v], c->sc->by_view.begin() + c->sc->by_view_off[(size_t)v + 1]);
        };
        add(job.ref_view);
        for (size_t g = 0; g < job.global.size(); ++g) add(job.global[g]);
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    }
    const size_t n_iter = by_view ? cand.size() : nf;
    for (size_t it = 0; it < n_iter; ++it) {
        const size_t i = by_view ? (size_t)cand[it] : it;
        Feature const& f = c->sc->features[i];
        bool use = by_view;
        if (tab) {
            use = (size_t)job.ref_view < G.nv && G.refs[(size_t)job.ref_view * nf + i];
            for (size_t g = 0; !use && g < job.global.size(); ++g) use = G.refs[(size_t)job.global[g] * nf + i] != 0;
        } else if (!by_view) {
            use = contains_view(c, f, job.ref_view);
            for (size_t g = 0; !use && g < job.global.size(); ++g)
                if (contains_view(c, f, job.global[g])) use = true;
        }
        if (!use) continue;
        V3 p = mk(f.pos[0], f.pos[1], f.pos[2]);
        if (!R.pointInFrustum(p)) continue;
        if (!in_box(p, st->aabbMin, st->aabbMax)) continue;
        ++job.n_seeds;
        V3 cp = xform(R.w2c, p);                                            /* worldToScreenScaled */
        float sx = dot3(L.proj, cp.v), sy = dot3(L.proj + 3, cp.v), sz = dot3(L.proj + 6, cp.v);
        int const x = (int)mround(sx / sz - 0.5f);
        int const y = (int)mround(sy / sz - 0.5f);
        if (x < 0 || y < 0 || x >= L.w || y >= L.h) continue;              /* the sampler's border test fails anyway */
        DevEntry e; e.job = job_index; e.xy = x | (y << 16);
        DevHyp h; h.depth = norm3(sub(p, R.pos())); h.dzI = 0.f; h.dzJ = 0.f; h.views = 0xFFFFFFFFu; h.views_hi = 0xFFFFFFFFu;
        job.seeds.push_back(e);
        job.seed_hyp.push_back(h);
    }
}

/* The job records of a batch: 13 KB each (128 global views), of which a job with the default 20 uses 3 -- a vector that does not
 * zero what fill_job is about to write (value-initialising 400 of them was half a millisecond of one thread per batch) */
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef DefaultInitAlloc<U> other; };
    DefaultInitAlloc() = default;
    template <class U> DefaultInitAlloc(const DefaultInitAlloc<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<DevJob, DefaultInitAlloc<DevJob> > JobVec;
/* the part of a job record that is in use: everything up to its last global view (the view records are the struct's tail) */
size_t job_prefix_bytes(int n_global) { return offsetof(DevJob, gv) + (size_t)std::min(std::max(n_global, 0), MI_MAX_GLOBAL) * sizeof(DevJobView); }

void fill_job(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, JobHost const& jh, DevJob& d) {
    HostView const& R = c->sc->views[jh.ref_view];
    HostLevel const& L = R.levels[st->scale];
    std::memset(&d, 0, job_prefix_bytes((int)jh.global.size()));
    d.ref_view = jh.ref_view; d.scale = st->scale; d.w = L.w; d.h = L.h;
    d.inv_a = L.invproj[0]; d.inv_c = L.invproj[2]; d.inv_b = L.invproj[4]; d.inv_d = L.invproj[5];
    const float* r = R.cam.rot;
    const float rt[9] = {r[0], r[3], r[6], r[1], r[4], r[7], r[2], r[5], r[8]};
    std::memcpy(d.rot_t, rt, sizeof(rt));
    std::memcpy(d.cam_pos, R.cam_pos, sizeof(d.cam_pos));
    d.w2c_z[0] = R.w2c[8]; d.w2c_z[1] = R.w2c[9]; d.w2c_z[2] = R.w2c[10]; d.w2c_z[3] = R.w2c[11];
    d.inv0_s = L.invproj[0];
    /* fault injection for tests (mi_dmrecon_debug_inject_footprint): the chosen reference view gets a negative pixel
     * footprint, the condition under which PatchSampler throws std::out_of_range (patch_sampler.cc:78-82) -- with
     * valid cameras it cannot be reached from outside */
    if (g_inject_footprint.load() == jh.ref_view) d.inv0_s = -d.inv0_s;
    d.n_global = (int)jh.global.size();
    for (size_t g = 0; g < jh.global.size(); ++g) {
        d.global_ids[g] = jh.global[g];
        HostView const& N = c->sc->views[jh.global[g]];
        DevJobView& J = d.gv[g];
        /* H = R_n R_ref^T K_s^-1, sc = R_n C_ref + t_n (DevJobView), in double, rounded once */
        const double Ki[9] = {L.invproj[0], 0.0, L.invproj[2], 0.0, L.invproj[4], L.invproj[5], 0.0, 0.0, 1.0};
        for (int i = 0; i < 3; ++i) {
            double mr[3];                                          /* row i of R_n R_ref^T */
            for (int k = 0; k < 3; ++k) mr[k] = (double)N.w2c[4 * i] * rt[k] + (double)N.w2c[4 * i + 1] * rt[3 + k] + (double)N.w2c[4 * i + 2] * rt[6 + k];
            for (int k = 0; k < 3; ++k) J.H[3 * i + k] = (float)(mr[0] * Ki[k] + mr[1] * Ki[3 + k] + mr[2] * Ki[6 + k]);
            J.sc[i] = (float)((double)N.w2c[4 * i] * R.cam_pos[0] + (double)N.w2c[4 * i + 1] * R.cam_pos[1]
                              + (double)N.w2c[4 * i + 2] * R.cam_pos[2] + (double)N.w2c[4 * i + 3]);
        }
        std::memcpy(J.w2c_z, N.w2c + 8, sizeof(J.w2c_z));
        J.inv0 = N.levels[0].invproj[0];
        J.maxl = (int)N.levels.size() - 1;
        J.view = jh.global[g];
        std::memcpy(J.cam_pos, N.cam_pos, sizeof(J.cam_pos));
    }
}

/* patches per wavefront of the throughput layout: a quad per patch, an octet for nrReconNeighbors > 4, a row of 16 lanes above 8 */
unsigned patches_per_wave(const mi_dmrecon_settings* st) { return st->nrReconNeighbors > 8 ? 4u : st->nrReconNeighbors > 4 ? 8u : (unsigned)MI_PATCHES_PER_WAVE; }

DevSettings dev_settings(const mi_dmrecon_settings* st) {
    DevSettings d;
    d.minNCC = st->minNCC; d.minParallax = st->minParallax; d.acceptNCC = st->acceptNCC;
    d.minRefineDiff = st->minRefineDiff; d.maxIterations = st->maxIterations; d.K = st->nrReconNeighbors;
    d.useColorScale = st->useColorScale;
    d.self_round = 0;
    d.ncc_stride = st->globalVSMax > 64 ? MI_MAX_GLOBAL : 64;
    d.seed_reopt = 0;
    return d;
}

/* Lays the per-pixel state maps of a batch out in the two map buffers and points the jobs at them. */
int alloc_maps(mi_dmrecon_ctx* c, std::vector<JobHost>& jobs, JobVec& dj, size_t& total_px, size_t n_list, int K) {
    const bool eight_views = K > 4, wide = K > 8;
    total_px = 0;
    for (size_t j = 0; j < jobs.size(); ++j) { jobs[j].pix_off = total_px; total_px += (size_t)jobs[j].w * jobs[j].h; }
    /* two state slots per pixel (dmrecon_types.h: DevJob): 2 x 7 floats; views, upd + views1, upd1 (+ views_hi,
     * views1_hi: view slots 4..7 of a set, nrReconNeighbors > 4) */
    const size_t n_imaps = eight_views ? 6 : 4;
    /* ONE sizing step for everything that goes with the pixels of the batch -- the maps AND the lists (n_list: the
     * longest list the call will hold, the seed list of a coarse scale can exceed the pixels): a later growth of the
     * lists would reallocate the maps the jobs already point at (DevBuf::reserve frees and allocates anew) */
    if (c->bs.ensure_pixels(std::max(total_px, n_list), n_imaps)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(maps) failed");
    float* base = c->bs.d_maps.p;
    float* base1 = base + 7 * total_px;
    uint32_t* ibase = c->bs.d_imaps.p;
    for (size_t j = 0; j < jobs.size(); ++j) {
        const size_t o = jobs[j].pix_off;
        dj[j].depth = base + o;
        dj[j].conf = base + total_px + o;
        dj[j].dz = base + 2 * total_px + 2 * o;
        dj[j].normal = base + 4 * total_px + 3 * o;
        dj[j].views = ibase + o;
        dj[j].upd = (int32_t*)(ibase + total_px + o);
        dj[j].depth1 = base1 + o;
        dj[j].conf1 = base1 + total_px + o;
        dj[j].dz1 = base1 + 2 * total_px + 2 * o;
        dj[j].normal1 = base1 + 4 * total_px + 3 * o;
        dj[j].views1 = ibase + 2 * total_px + o;
        dj[j].upd1 = (int32_t*)(ibase + 3 * total_px + o);
        dj[j].views_hi = eight_views ? ibase + 4 * total_px + o : nullptr;
        dj[j].views1_hi = eight_views ? ibase + 5 * total_px + o : nullptr;
        dj[j].views_x = nullptr; dj[j].results_x = nullptr; dj[j].hyp_x = nullptr;
    }
    if (wide) {
        /* sixteen view slots: slots 8..15 of the sets, two words per pixel | per entry of a round's list (the batch's) | per
         * explicit hypothesis (only the parity hook has any: it points hyp_x at the third part) */
        const size_t cap = std::max(total_px, n_list);
        if (c->bs.d_xviews.reserve(2 * total_px + 4 * cap)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(view sets) failed");
        uint32_t* xb = c->bs.d_xviews.p;
        for (size_t j = 0; j < jobs.size(); ++j) { dj[j].views_x = xb + 2 * jobs[j].pix_off; dj[j].results_x = xb + 2 * total_px; }
        HIP_TRY(hipMemsetAsync(xb, 0xFF, (2 * total_px + 4 * cap) * sizeof(uint32_t), c->stream));
    }
    /* slot 1 is only ever read where its stamp says so: the stamps (0xFF.. = -1) are all it needs */
    HIP_TRY(hipMemsetAsync(c->bs.d_maps.p, 0, total_px * 7 * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->bs.d_imaps.p, 0xFF, total_px * n_imaps * sizeof(uint32_t), c->stream));
    return 0;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

/* =========================================================================== */
extern "C" {

int mi_dmrecon_abi_version(void) { return MI_DMRECON_ABI_VERSION; }

int mi_dmrecon_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mi_dmrecon_last_error(void) { return g_err.c_str(); }

int mi_dmrecon_local_view_channels(int32_t nrReconNeighbors) { return nrReconNeighbors > 8 ? 16 : nrReconNeighbors > 4 ? 8 : 4; }

void mi_dmrecon_settings_default(mi_dmrecon_settings* s) {          /* libs/dmrecon/settings.h:25-51 */
    s->filterWidth = 5; s->minNCC = 0.3f; s->minParallax = 10.f; s->acceptNCC = 0.6f; s->minRefineDiff = 0.001f;
    s->maxIterations = 20; s->nrReconNeighbors = 4; s->globalVSMax = 20; s->scale = 0; s->useColorScale = 1;
    for (int i = 0; i < 3; ++i) {
        s->aabbMin[i] = -std::numeric_limits<float>::max();
        s->aabbMax[i] = std::numeric_limits<float>::max();
    }
}

static int create_streams(mi_dmrecon_ctx* c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    /* The second stream carries the maps of finished views back while the front kernel still runs on the first: it must
     * not share a hardware queue with it.  The runtime deals the streams of one priority over a handful of queues (four by
     * default): with four forked contexts -- eight streams -- a context's two streams could land on the same one, and its
     * copies then waited for the kernel they were meant to run next to (a 400-view batch: 31 ms = its 830 MB of maps over
     * PCIe, in the two regions out of five that such a context led).  Streams of another priority have queues of their own. */
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { prio_least = prio_greatest = 0; (void)hipGetLastError(); }
    if (prio_greatest != prio_least) HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_greatest));
    else HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    /* compute units of this device (a partitioned GPU has fewer than 256): what a front launch with teams may occupy */
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->n_cus = cus;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
    else (void)hipGetLastError();
    return 0;
}

/* The runtime gives a stream its hardware queue when the stream is first used -- tens of milliseconds, which used to land
 * in whichever call first led a batch on this context (the second stream: in its first front phase): used once here. */
static int warm_streams(mi_dmrecon_ctx* c) {
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    return 0;
}

int mi_dmrecon_ctx_create(int device, mi_dmrecon_ctx** out) {
    if (!out) return fail(MI_DMRECON_EINVAL, "null out pointer");
    int n = mi_dmrecon_device_count();
    if (n <= 0) return fail(MI_DMRECON_EDEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(MI_DMRECON_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    mi_dmrecon_ctx* c = new mi_dmrecon_ctx;
    c->device = device;
    c->sc = std::make_shared<SceneStore>();
    c->sc->device = device;
    if (int rc = create_streams(c)) return rc;
    /* sRGB -> linear table: the formula documented at mvs_tools.cc:22-29 evaluated in double and
     * rounded to float reproduces the literal table at :30-93 bit for bit (tests/test_host_logic.py). */
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        double x = i / 255.0;
        lut[i] = (float)((i <= 0.04045 * 255.0) ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
    }
    lut[255] = 1.0f;
    HIP_TRY(hipMalloc((void**)&c->sc->d_lut, sizeof(lut)));
    HIP_TRY(hipMemcpy(c->sc->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&c->d_counters, sizeof(DevCounters)));
    if (int rc = warm_streams(c)) return rc;
    *out = c;
    return 0;
}

void mi_dmrecon_ctx_destroy(mi_dmrecon_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)wait_stream(c->stream);
    if (c->stream2) (void)wait_stream(c->stream2);
    c->bs.release();
    c->d_stage.release(); c->d_stage2.release();
    if (c->d_counters) (void)hipFree(c->d_counters);
    (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;                                 /* the scene store goes with its last owner */
}

int mi_dmrecon_ctx_fork(mi_dmrecon_ctx* parent, mi_dmrecon_ctx** out) {
    if (!parent || !out) return fail(MI_DMRECON_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(parent->device));
    mi_dmrecon_ctx* c = new mi_dmrecon_ctx;
    c->device = parent->device;
    c->sc = parent->sc;
    if (int rc = create_streams(c)) return rc;
    HIP_TRY(hipMalloc((void**)&c->d_counters, sizeof(DevCounters)));
    if (int rc = warm_streams(c)) return rc;
    *out = c;
    return 0;
}

/* Page-locked host memory for the output maps: device->host copies into it run at full PCIe rate
 * and without the staging copy that pageable memory needs. */
void* mi_dmrecon_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { fail(MI_DMRECON_EDEVICE, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void mi_dmrecon_host_free(void* p) { if (p) (void)hipHostFree(p); }

void* mi_dmrecon_ctx_stream(mi_dmrecon_ctx* c) { return c ? (void*)c->stream : nullptr; }

/* The host half of a view: camera, world-to-camera matrix and the calibration of every pyramid level (no pixels).
 * Returns the texels over all levels. */
static size_t host_view_set_camera(HostView& v, const mi_dmrecon_camera* cam, int32_t width, int32_t height) {
    v.cam = *cam;
    v.valid = cam->flen != 0.f;                                   /* CameraInfo::is_valid (View::is_camera_valid) */
    const float* rot = cam->rot; const float* t = cam->trans;
    v.cam_pos[0] = -rot[0] * t[0] - rot[3] * t[1] - rot[6] * t[2];      /* camera.cc:34-39 */
    v.cam_pos[1] = -rot[1] * t[0] - rot[4] * t[1] - rot[7] * t[2];
    v.cam_pos[2] = -rot[2] * t[0] - rot[5] * t[1] - rot[8] * t[2];
    float* m = v.w2c;                                                   /* camera.cc:63-69 */
    m[0] = rot[0]; m[1] = rot[1]; m[2] = rot[2]; m[3] = t[0];
    m[4] = rot[3]; m[5] = rot[4]; m[6] = rot[5]; m[7] = t[1];
    m[8] = rot[6]; m[9] = rot[7]; m[10] = rot[8]; m[11] = t[2];
    m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
    /* buildPyramid, image_pyramid.cc:21-53 */
    v.levels.clear();
    mi_dmrecon_camera pc = *cam;
    int cw = width, ch = height;
    size_t off = 0;
    HostLevel l0; l0.w = cw; l0.h = ch; l0.tex_off = 0;
    calibration(pc, (float)cw, (float)ch, l0.proj, l0.invproj);
    v.levels.push_back(l0);
    /* (a level's slot holds its texels row by row AND -- four times that, behind the levels -- its footprint records; the slot
     * is padded to whole 8 x 8 tiles so that a block-linear arrangement of the records fits as well: the -DMI_TILED_QUADS
     * experiment of the device code.  < 1 % of a view's memory.) */
    auto slot = [](int w, int h) { return (size_t)((w + 7) & ~7) * (size_t)((h + 7) & ~7); };
    off += slot(cw, ch);
    while (std::min(cw, ch) >= 30 && v.levels.size() < MI_MAX_LEVELS) {
        if (cw % 2 == 1) pc.ppoint[0] = pc.ppoint[0] * float(cw) / float(cw + 1);
        if (ch % 2 == 1) pc.ppoint[1] = pc.ppoint[1] * float(ch) / float(ch + 1);
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        HostLevel l; l.w = cw; l.h = ch; l.tex_off = (uint32_t)off;
        calibration(pc, (float)cw, (float)ch, l.proj, l.invproj);
        v.levels.push_back(l);
        off += slot(cw, ch);
    }
    return off;
}

static int set_view_impl(mi_dmrecon_ctx* c, int32_t view_id, const mi_dmrecon_camera* cam, int32_t width,
                         int32_t height, int32_t channels, const uint8_t* pixels, bool async) {
    if (!c || !cam) return fail(MI_DMRECON_EINVAL, "null argument");
    if (view_id < 0 || view_id >= (1 << 20)) return fail(MI_DMRECON_EINVAL, "bad view id %d", view_id);
    if (width < 2 || height < 2 || width > 65535 || height > 65535) return fail(MI_DMRECON_EINVAL, "bad image size %dx%d", width, height);
    if (pixels && (channels < 1 || channels > 4)) return fail(MI_DMRECON_EINVAL, "Image with invalid number of channels");
    HIP_TRY(hipSetDevice(c->device));
    if ((size_t)view_id >= c->sc->views.size()) c->sc->views.resize(view_id + 1);
    HostView& v = c->sc->views[view_id];
    if (v.d_img) { (void)hipFree(v.d_img); v.d_img = nullptr; }
    const size_t off = host_view_set_camera(v, cam, width, height);
    if (!pixels) {
        /* camera and image size only (mi_dmrecon.h): a candidate of the view selections whose image could not be loaded */
        v.n_texels = 0;
        c->sc->views_dirty = true;
        c->sc->geom.built = false;
        return 0;
    }
    v.n_texels = off;
    /* RGBA8 levels, then (16-byte aligned) the same levels as footprint elements (column pairs, 8 bytes per texel position: a
     * sample's 2 x 2 footprint is one 16-byte gather over two of them; the last gather of the last level ends 8 bytes past its
     * elements, hence the pad): 4 + 8 bytes per texel */
    v.quad_off = (off + 3) & ~(size_t)3;
    HIP_TRY(hipMalloc((void**)&v.d_img, (v.quad_off + (size_t)mi_quad_words() * off + 4) * sizeof(uint32_t)));
    /* ensureImages, image_pyramid.cc:55-95: upload, strip alpha / expand grey, then the Gaussian levels */
    const size_t nbytes = (size_t)width * height * channels;
    /* two device staging buffers used alternately: the copy of view i+1 (from pinned memory) can start
     * while the pack/pyramid kernels of view i still read the other one */
    DevBuf<uint8_t>& stage = (c->stage_flip ^= 1) ? c->d_stage : c->d_stage2;
    if (stage.cap < nbytes) {
        HIP_TRY(wait_stream(c->stream));          /* the buffer may still be in use */
        if (stage.reserve(nbytes)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(stage) failed");
    }
    HIP_TRY(hipMemcpyAsync(stage.p, pixels, nbytes, hipMemcpyHostToDevice, c->stream));
    mi_launch_pack_rgba(c->stream, stage.p, v.d_img, width * height, channels);
    const float w1 = std::exp(-0.5f / (2.0f * 1.f)), w2 = std::exp(-2.5f / (2.0f * 1.f)), w3 = std::exp(-4.5f / (2.0f * 1.f));
    for (size_t l = 1; l < v.levels.size(); ++l) {
        HostLevel const& a = v.levels[l - 1]; HostLevel const& b = v.levels[l];
        mi_launch_pyramid(c->stream, v.d_img + a.tex_off, v.d_img + b.tex_off, a.w, a.h, b.w, b.h, w1, w2, w3);
    }
    for (size_t l = 0; l < v.levels.size(); ++l) {
        HostLevel const& a = v.levels[l];
        mi_launch_quadify(c->stream, v.d_img + a.tex_off, v.d_img + v.quad_off + (size_t)mi_quad_words() * (size_t)a.tex_off, a.w, a.h);
    }
    HIP_TRY(hipGetLastError());
    if (!async) HIP_TRY(wait_stream(c->stream));
    c->sc->views_dirty = true;
    c->sc->geom.built = false;
    return 0;
}

int mi_dmrecon_set_view(mi_dmrecon_ctx* c, int32_t view_id, const mi_dmrecon_camera* cam, int32_t width,
                        int32_t height, int32_t channels, const uint8_t* pixels) {
    return set_view_impl(c, view_id, cam, width, height, channels, pixels, false);
}

int mi_dmrecon_set_view_async(mi_dmrecon_ctx* c, int32_t view_id, const mi_dmrecon_camera* cam, int32_t width,
                              int32_t height, int32_t channels, const uint8_t* pixels) {
    return set_view_impl(c, view_id, cam, width, height, channels, pixels, true);
}

int mi_dmrecon_sync(mi_dmrecon_ctx* c) {
    if (!c) return fail(MI_DMRECON_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(wait_stream(c->stream));
    return 0;
}

int mi_dmrecon_evict_view(mi_dmrecon_ctx* c, int32_t view_id) {
    if (!c || view_id < 0 || (size_t)view_id >= c->sc->views.size()) return fail(MI_DMRECON_EINVAL, "bad view id");
    HostView& v = c->sc->views[view_id];
    if (v.d_img) { (void)hipFree(v.d_img); v.d_img = nullptr; }
    v.valid = false; v.levels.clear();
    c->sc->views_dirty = true;
    c->sc->geom.built = false;
    return 0;
}

static int mi_dmrecon_set_features_impl(mi_dmrecon_ctx* c, int32_t n, const float* pos, const int32_t* off, const int32_t* ids) {
    if (!c || n < 0 || (n > 0 && (!pos || !off || !ids))) return fail(MI_DMRECON_EINVAL, "null argument");
    c->sc->features.resize(n);
    for (int i = 0; i < n; ++i) {
        Feature& f = c->sc->features[i];
        f.pos[0] = pos[3 * i]; f.pos[1] = pos[3 * i + 1]; f.pos[2] = pos[3 * i + 2];
        f.ref_begin = off[i]; f.ref_end = off[i + 1];
    }
    c->sc->feat_refs.assign(ids, ids + (n ? off[n] : 0));
    build_features_by_view(*c->sc);
    c->sc->geom.built = false;
    return 0;
}

int mi_dmrecon_num_levels(mi_dmrecon_ctx* c, int32_t view_id) {
    if (!c || view_id < 0 || (size_t)view_id >= c->sc->views.size() || !c->sc->views[view_id].d_img)
        return fail(MI_DMRECON_EINVAL, "unknown view %d", view_id);
    return (int)c->sc->views[view_id].levels.size();
}

int mi_dmrecon_level_size(mi_dmrecon_ctx* c, int32_t view_id, int32_t level, int32_t* w, int32_t* h) {
    int n = mi_dmrecon_num_levels(c, view_id);
    if (n < 0) return n;
    if (level < 0 || level >= n) return fail(MI_DMRECON_EINVAL, "level %d out of range", level);
    if (w) *w = c->sc->views[view_id].levels[level].w;
    if (h) *h = c->sc->views[view_id].levels[level].h;
    return 0;
}

int mi_dmrecon_get_level(mi_dmrecon_ctx* c, int32_t view_id, int32_t level, uint8_t* rgb, float* proj, float* invproj) {
    int n = mi_dmrecon_num_levels(c, view_id);
    if (n < 0) return n;
    if (level < 0 || level >= n) return fail(MI_DMRECON_EINVAL, "level %d out of range", level);
    HIP_TRY(hipSetDevice(c->device));
    HostView const& v = c->sc->views[view_id];
    HostLevel const& L = v.levels[level];
    if (proj) std::memcpy(proj, L.proj, sizeof(L.proj));
    if (invproj) std::memcpy(invproj, L.invproj, sizeof(L.invproj));
    if (rgb) {
        const size_t np = (size_t)L.w * L.h;
        if (c->d_stage.reserve(np * 3)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(stage) failed");
        mi_launch_unpack_rgb(c->stream, v.d_img + L.tex_off, c->d_stage.p, (int)np);
        HIP_TRY(hipMemcpyAsync(rgb, c->d_stage.p, np * 3, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(wait_stream(c->stream));
    }
    return 0;
}

static int mi_dmrecon_global_view_selection_impl(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view,
                                     int32_t* ids_out, int32_t* n_out) {
    if (!c || !ids_out || !n_out) return fail(MI_DMRECON_EINVAL, "null argument");
    int rc = check_settings(st);
    if (rc) return rc;
    std::vector<int> g;
    rc = 1;
    if (gvs_device_wanted(c, 1)) {
        HIP_TRY(hipSetDevice(c->device));
        std::vector<std::vector<int> > gl(1); std::vector<int> vrc(1, 0); std::vector<std::string> err(1);
        rc = plan_global_views_device(c, st, 1, &ref_view, gl, vrc, err);
        if (rc < 0) return rc;
        if (rc == 0) { if (vrc[0]) { g_err = err[0]; return vrc[0]; } g.swap(gl[0]); }
    }
    if (rc == 1) rc = plan_global_views(c, st, ref_view, g);
    if (rc) return rc;
    for (size_t i = 0; i < g.size(); ++i) ids_out[i] = g[i];
    *n_out = (int)g.size();
    return 0;
}

/* ---- one batch of reference views: what mvs::DMRecon::start() does for each of them (dmrecon.cc:89-172) ---------- */
namespace {

/* the two words of a DevJob the device writes and the host polls, copied back as a strided 8-byte column */
struct JobDyn { int32_t flags; uint32_t n_filled; uint32_t list; /* the view's list of the round reported (k_round_report), else stale */ };

/* hipEvent pairs around the timed launches of a call, recorded on the stream the launch goes to */
struct EventLog {
    enum { BULK = 0, SWEEP = 1, TAIL = 2, FRONT = 3, LAT = 4 /* a bulk launch in the latency layout */ };
    struct Item { size_t first; int kind; unsigned work; bool ok; };
    mi_dmrecon_ctx* c = nullptr;
    size_t n_ev = 0;
    std::vector<Item> items;
    hipEvent_t get(size_t i) {
        while (c->bs.events.size() <= i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; c->bs.events.push_back(e); }
        return c->bs.events[i];
    }
    void begin(hipStream_t S, int kind, unsigned work) {
        Item it; it.first = n_ev; it.kind = kind; it.work = work;
        hipEvent_t e0 = get(n_ev), e1 = get(n_ev + 1);
        it.ok = e0 && e1 && hipEventRecord(e0, S) == hipSuccess;
        items.push_back(it); n_ev += 2;
    }
    void end(hipStream_t S) { Item& it = items.back(); if (it.ok) it.ok = hipEventRecord(c->bs.events[it.first + 1], S) == hipSuccess; }
    bool ms(const Item& it, float& out) const {
        return it.ok && hipEventElapsedTime(&out, c->bs.events[it.first], c->bs.events[it.first + 1]) == hipSuccess;
    }
};

struct BatchRun {
    /* the call */
    mi_dmrecon_ctx* c; const mi_dmrecon_settings* st; int n_refs; const int32_t* ref_views;
    mi_dmrecon_maps* maps; mi_dmrecon_progress* progress; int32_t* status_out; mi_dmrecon_stats* stats;
    const MiDeviceApi* D; DevSettings ds; hipStream_t S;
    bool trace; double t_begin, t_mark;
    /* the plan: one job per reference view that got through the host planning */
    std::vector<int> view_rc, job_of, ref_of_job;
    std::vector<std::string> plan_err;
    std::vector<JobHost> jobs; JobVec dj;
    int nj = 0, n_alive = 0, max_tiles = 0;
    size_t total_px = 0, work_cap = 0, n_seed_feats = 0;
    size_t up_jobs = 0, up_keyoff = 0, up_seeds = 0, up_hyps = 0;   /* offsets into the pinned upload staging (BatchScratch::h_up) */
    size_t job_bytes = sizeof(DevJob);         /* the part of a job record that is uploaded (DevJob: up to the last global view in use) */
    size_t n_seeds_total = 0;                  /* seeds of all views of the batch (they go from the plans straight into the pinned staging) */
    std::vector<unsigned> keyoff;
    /* the rounds */
    EventLog ev;
    int round = 1;
    bool done = false, truncated = false, have_handover = false;
    unsigned tail_known = 0;
    DevCounters hc;
    int64_t n_launch = 0, n_tail_launch = 0, n_tail_timed = 0;
    /* list + results of the last executed tail round, and their ping-pong partners */
    DevEntry* wcur = nullptr; DevEntry* wnext = nullptr; DevResult* rcur = nullptr; DevResult* rnext = nullptr;
    /* the front kernel (phase C) */
    std::vector<char> streamed;                /* views whose maps went back to the host while the front kernel still ran */
    int n_streamed = 0, n_streamed_early = 0;
    int stream_view(int j);
    /* large batches: the maps go back as a snapshot taken at the hand-over (copied while the front kernel runs) plus the
     * list of the pixels the front changed afterwards (front_rounds) */
    bool sparse = false, sparse_overflow = false;
    int sparse_r0 = 0;                         /* the first round whose writes the snapshot may have missed */
    unsigned sparse_stride = 6, sparse_cap = 0, sparse_done_end = 0;   /* words per record, records the list holds, records scattered so far */
    std::vector<char> snapped;                 /* views whose snapshot is on its way */
    std::vector<int> emit_order; size_t emit_seen = 0;
    int snapshot_views();
    int emit_view(int j);
    void scatter_emitted(bool all);
    bool ran_front = false; int front_first_round = 0, front_team = 1, front_team_max = 1, front_fallbacks = 0;
    std::vector<unsigned> view_filled;       /* ... and the pixels the view had filled by then */
    std::vector<unsigned> view_list;         /* entries of every view's list in the last host-visible round read back (0: not known) */
    std::vector<unsigned> front_order;       /* one workgroup per view: the views in the order their workgroups start (FrontArgs::job_order) */
    std::vector<unsigned> front_map; unsigned front_grid = 0;   /* teams: what every block of the front launch is (FrontArgs::block_map) */
    int seed_mode = 2;                         /* MI_DMRECON_SEED_REOPT: 0 every seed propagates at once, 1 re-optimisation as round 1, 2 (default) in the seed launch */
    unsigned handover = MI_VIEW_HANDOVER;      /* k_generate: a view's own list size below which it leaves the throughput layout */
    bool host_rounds_only = false;             /* diagnostic: every round host-visible (MI_DMRECON_HOST_ROUNDS) */
    int n_lat_rounds = 0;                      /* host-visible rounds that had entries in the latency layout */
    std::vector<unsigned> front_stats;
    const ActiveCall* active_call = nullptr;   /* reconstruct calls in progress on this GPU */

    void mark(const char* what) {
        if (!trace) return;
        const double t = now_ms();
        fprintf(stderr, "[mi_dmrecon] phase %-22s %8.3f ms\n", what, t - t_mark);
        t_mark = t;
    }
    JobDyn* dyn_of(int slot) { return (JobDyn*)c->bs.h_dyn + (size_t)slot * nj; }
    hipError_t read_dyn(int slot) {
        return hipMemcpy2DAsync(dyn_of(slot), sizeof(JobDyn), (const char*)c->bs.d_jobs.p + offsetof(DevJob, flags), sizeof(DevJob),
                                2 * sizeof(uint32_t), (size_t)nj, hipMemcpyDeviceToHost, S);
    }
    void plan_front_team();
    int plan();
    int upload();
    int seed_round();
    int poll_views(const JobDyn* dyn, unsigned queue_size);
    int bulk_rounds(bool& to_tail);
    int tail_rounds(bool& to_front);
    int front_rounds();
    int download();
    void fill_stats();
    int outcome();
};

/* ---- host planning: global view selection and seeds, one plan per reference view.
 * Returns 0 to go on, a negative code when the call is over (g_err set). */
int BatchRun::plan() {
    view_rc.assign(n_refs, 0); job_of.assign(n_refs, -1); plan_err.assign(n_refs, std::string());
    for (int i = 0; i < n_refs; ++i) {
        if (progress) progress[i].start_time = (uint64_t)std::time(nullptr);
        if (progress && progress[i].cancelled) { view_rc[i] = MI_DMRECON_ECANCELLED; progress[i].status = MI_RECON_CANCELLED; }   /* dmrecon.cc:101-105 */
        else if (progress) progress[i].status = MI_RECON_GLOBALVS;
    }
    std::vector<JobHost> plans(n_refs);
    const int n_threads = std::max(1, std::min(n_refs, host_threads_cap()));
    bool gvs_done = false;
    if (gvs_device_wanted(c, n_refs)) {
        std::vector<std::vector<int> > gl(n_refs);
        const int r = plan_global_views_device(c, st, n_refs, ref_views, gl, view_rc, plan_err);
        if (r < 0) return r;
        if (r == 0) {
            gvs_done = true;
            for (int i = 0; i < n_refs; ++i) {
                if (view_rc[i]) continue;
                plans[i].ref_view = ref_views[i];
                plans[i].global.swap(gl[i]);
                if (plans[i].global.empty()) { view_rc[i] = fail(MI_DMRECON_EGVS, "Global View Selection failed"); plan_err[i] = g_err; }
            }
        }
    }
    /* A reference view that is listed several times in one call (a merged batch whose callers ask for the same views; a
     * caller that reconstructs a scene's views repeatedly) is planned ONCE: its global view set and its seeds are functions of
     * the scene, the view and the settings alone (dmrecon.cc:178-331 reads nothing else), so the later entries copy the first
     * one's plan (same_as[i]: the first entry with this view, i itself for that one). */
    std::vector<int> same_as((size_t)n_refs);
    {
        std::unordered_map<int32_t, int> first;
        for (int i = 0; i < n_refs; ++i) same_as[i] = first.emplace(ref_views[i], i).first->second;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int i = 0; i < n_refs; ++i) {
        if (gvs_done || view_rc[i] || same_as[i] != i) continue;
        plans[i].ref_view = ref_views[i];
        int r = plan_global_views(c, st, ref_views[i], plans[i].global);
        if (r == 0 && plans[i].global.empty()) r = fail(MI_DMRECON_EGVS, "Global View Selection failed");
        view_rc[i] = r;
        if (r) plan_err[i] = g_err;
    }
    for (int i = 0; i < n_refs && !gvs_done; ++i) {
        const int k = same_as[i];
        if (k == i || view_rc[i]) continue;                     /* (cancelled before it began: stays cancelled) */
        plans[i].ref_view = ref_views[i];
        if (view_rc[k] == MI_DMRECON_ECANCELLED) {              /* the first entry was cancelled, this one is not: its own selection */
            int r = plan_global_views(c, st, ref_views[i], plans[i].global);
            if (r == 0 && plans[i].global.empty()) r = fail(MI_DMRECON_EGVS, "Global View Selection failed");
            view_rc[i] = r;
            if (r) plan_err[i] = g_err;
            same_as[i] = i;
            continue;
        }
        plans[i].global = plans[k].global; view_rc[i] = view_rc[k]; plan_err[i] = plan_err[k];
    }
    /* a selected view that was registered without pixels (mi_dmrecon_set_view with a null image: its image could not be loaded
     * by the caller): the reference fails here, when it loads the selected views (dmrecon.cc:236-240) */
    for (int i = 0; i < n_refs; ++i) {
        if (view_rc[i]) continue;
        for (int g : plans[i].global)
            if (c->device >= 0 && g >= 0 && (size_t)g < c->sc->views.size() && !c->sc->views[g].d_img) {
                view_rc[i] = fail(MI_DMRECON_ENOIMAGE, "view %d, selected as a neighbour of view %d, has no image", g, ref_views[i]);
                plan_err[i] = g_err;
                break;
            }
    }
    mark("global view selection");
    const double t_gvs_done = now_ms();
    if (stats) { stats->gvs_on_device = gvs_done ? 1 : 0; stats->ms_plan_gvs = t_gvs_done - t_begin; }
    for (int i = 0; i < n_refs; ++i) {
        if (status_out) status_out[i] = view_rc[i];
        if (view_rc[i]) continue;
        HostLevel const& L = c->sc->views[ref_views[i]].levels[st->scale];
        plans[i].w = L.w; plans[i].h = L.h;
        job_of[i] = (int)jobs.size();
        ref_of_job.push_back(i);
        jobs.push_back(std::move(plans[i]));
    }
    if (jobs.empty()) {
        /* no view got through: the first failing view's own code and message (a single-view call behaves like
         * DMRecon's constructor / start(): the exception is the call's outcome), a cancellation only if all were */
        for (int i = 0; i < n_refs; ++i)
            if (view_rc[i] != MI_DMRECON_ECANCELLED) { g_err = plan_err[i].empty() ? "Global View Selection failed" : plan_err[i]; return view_rc[i]; }
        return fail(MI_DMRECON_ECANCELLED, "cancelled");
    }
    for (int i = 0; progress && i < n_refs; ++i) if (view_rc[i] == 0) progress[i].status = MI_RECON_FEATURES;
    /* (seeds: once per distinct reference view of the call, as above -- a copy only differs in the job its entries name) */
    std::vector<int> job_same((size_t)jobs.size());
    for (int j = 0; j < (int)jobs.size(); ++j) {
        const int k = same_as[ref_of_job[j]];
        job_same[j] = (k != ref_of_job[j] && job_of[k] >= 0 && job_of[k] < j) ? job_of[k] : j;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int j = 0; j < (int)jobs.size(); ++j) if (job_same[j] == j) plan_seeds(c, st, jobs[j], j);
#pragma omp parallel for schedule(static) num_threads(n_threads) if (jobs.size() >= 64)
    for (int j = 0; j < (int)jobs.size(); ++j) {
        const int k = job_same[j];
        if (k == j) continue;
        jobs[j].seeds = jobs[k].seeds; jobs[j].seed_hyp = jobs[k].seed_hyp; jobs[j].n_seeds = jobs[k].n_seeds;
        for (DevEntry& e : jobs[j].seeds) e.job = j;
    }
    mark("seed planning");
    if (stats) stats->ms_plan_seeds = now_ms() - t_gvs_done;
    return 0;
}

/* ---- job table, state maps, seeds and work-list buffers on the device (all copies asynchronous) */
int BatchRun::upload() {
    int rc = sync_views(c);
    if (rc) return rc;
    nj = (int)jobs.size();
    dj.resize(nj);
    /* (a merged batch of 400 views: 8 000 per-view geometries and 13 MB of seeds -- by a few threads, the seeds from the
     * views' plans straight into the pinned staging; it was 9 ms of one thread) */
    const int n_threads = std::max(1, std::min(std::min(nj / 4, host_threads_cap()), 16));
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
    for (int j = 0; j < nj; ++j) fill_job(c, st, jobs[j], dj[j]);
    std::vector<size_t> seed_off((size_t)nj + 1, 0);
    for (int j = 0; j < nj; ++j) {
        const int tx = (jobs[j].w + MI_GEN_TILE_W - 1) / MI_GEN_TILE_W, ty = (jobs[j].h + MI_GEN_TILE_H - 1) / MI_GEN_TILE_H;
        max_tiles = std::max(max_tiles, tx * ty);
        seed_off[(size_t)j + 1] = seed_off[j] + jobs[j].seeds.size();
        n_seed_feats += jobs[j].n_seeds;
    }
    n_seeds_total = seed_off[nj];
    mark("  upload: job records");
    rc = alloc_maps(c, jobs, dj, total_px, n_seeds_total, st->nrReconNeighbors);
    if (rc) return rc;
    mark("  upload: state maps");
    work_cap = std::max(total_px, n_seeds_total);
    if (c->bs.d_jobs.reserve(nj)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(jobs) failed");
    keyoff.resize(nj);
    for (int j = 0; j < nj; ++j) keyoff[j] = (unsigned)jobs[j].pix_off;
    {
        /* everything the call uploads, through ONE page-locked staging buffer: asynchronous for real */
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        /* of a job record only the part its global views use goes up (the records of the views are its tail: DevJob), packed;
         * a kernel spreads the records to their places (mi_launch_unpack_jobs) */
        int max_global = 1;
        for (int j = 0; j < nj; ++j) max_global = std::max(max_global, (int)dj[j].n_global);
        job_bytes = job_prefix_bytes(max_global);
        up_jobs = 0; up_keyoff = al(up_jobs + (size_t)nj * job_bytes); up_seeds = al(up_keyoff + (size_t)nj * sizeof(unsigned));
        up_hyps = al(up_seeds + n_seeds_total * sizeof(DevEntry));
        const size_t need = al(up_hyps + n_seeds_total * sizeof(DevHyp));
        if (c->bs.h_up_cap < need) {
            if (c->bs.h_up) (void)hipHostFree(c->bs.h_up);
            c->bs.h_up = nullptr; c->bs.h_up_cap = 0;
            if (hipHostMalloc((void**)&c->bs.h_up, 2 * need, hipHostMallocDefault) != hipSuccess)
                return fail(MI_DMRECON_EDEVICE, "hipHostMalloc(upload staging) failed");
            c->bs.h_up_cap = 2 * need;
        }
        std::memcpy(c->bs.h_up + up_keyoff, keyoff.data(), (size_t)nj * sizeof(unsigned));
        uint8_t* const hu = c->bs.h_up;
        const size_t jb = job_bytes;
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
        for (int j = 0; j < nj; ++j) {
            std::memcpy(hu + up_jobs + (size_t)j * jb, &dj[j], job_prefix_bytes(dj[j].n_global));
            const size_t n = jobs[j].seeds.size();
            if (n) {
                std::memcpy(hu + up_seeds + seed_off[j] * sizeof(DevEntry), jobs[j].seeds.data(), n * sizeof(DevEntry));
                std::memcpy(hu + up_hyps + seed_off[j] * sizeof(DevHyp), jobs[j].seed_hyp.data(), n * sizeof(DevHyp));
            }
        }
    }
    mark("  upload: staging");
    if (c->bs.d_jobs_packed.reserve(((size_t)nj * job_bytes + 3) / 4)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(jobs) failed");
    HIP_TRY(hipMemcpyAsync(c->bs.d_jobs_packed.p, c->bs.h_up + up_jobs, (size_t)nj * job_bytes, hipMemcpyHostToDevice, S));
    mi_launch_unpack_jobs(S, c->bs.d_jobs_packed.p, (unsigned)(job_bytes / 4), c->bs.d_jobs.p, nj);
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), S));
    if (c->bs.d_hyp.reserve(std::max<size_t>(n_seeds_total, 1)) || c->bs.d_keyoff.reserve(nj)
        || c->bs.d_round_work.reserve(MI_MAX_ROUNDS) || c->bs.d_round_work_t.reserve(MI_MAX_ROUNDS) || c->bs.d_round_items.reserve(MI_MAX_ROUNDS)
        || c->bs.d_follow_cnt.reserve(MI_FOLLOW_LISTS * MI_MAX_ROUNDS) || c->bs.d_view.reserve(4 * (size_t)nj)
        || c->bs.d_front.reserve(7 * (size_t)nj) || c->bs.d_front_resume.reserve(2 * (size_t)nj))
        return fail(MI_DMRECON_EDEVICE, "hipMalloc(work lists) failed");
    HIP_TRY(hipMemsetAsync(c->bs.d_round_work.p, 0, MI_MAX_ROUNDS * sizeof(unsigned), S));
    HIP_TRY(hipMemsetAsync(c->bs.d_round_work_t.p, 0, MI_MAX_ROUNDS * sizeof(unsigned), S));
    HIP_TRY(hipMemsetAsync(c->bs.d_round_items.p, 0, MI_MAX_ROUNDS * sizeof(unsigned), S));
    /* MI_DMRECON_VIEW_HANDOVER=<entries> (read per call): a view's own list size below which it leaves the throughput
     * layout for good (k_generate decides, per view, on the device); 0 = never, 1000000000 = from the first round on */
    if (const char* e = std::getenv("MI_DMRECON_VIEW_HANDOVER")) handover = (unsigned)std::max(0L, std::atol(e));
    host_rounds_only = [] { const char* e = std::getenv("MI_DMRECON_HOST_ROUNDS"); return e && std::atoi(e) != 0; }();
    /* sixteen view slots (nrReconNeighbors > 8) exist in the throughput layout only: the views never hand over, every round is a
     * host-visible one (bulk_rounds: one launch of the general kernel per round) */
    if (st->nrReconNeighbors > 8) handover = 0;
    /* per-view counts of "the round before round 1": none yet -- unless every view starts in the latency layout */
    HIP_TRY(hipMemsetAsync(c->bs.d_view.p, 0, 4 * (size_t)nj * sizeof(unsigned), S));
    if (handover < 1000000000u) HIP_TRY(hipMemsetAsync(c->bs.d_view.p, 0xFF, (size_t)nj * sizeof(unsigned), S));
    HIP_TRY(hipMemsetAsync(c->bs.d_follow_cnt.p, 0, MI_FOLLOW_LISTS * MI_MAX_ROUNDS * sizeof(unsigned), S));
    HIP_TRY(hipMemcpyAsync(c->bs.d_keyoff.p, c->bs.h_up + up_keyoff, nj * sizeof(unsigned), hipMemcpyHostToDevice, S));
    if (!c->bs.h_poll) {
        /* (mapped: written by k_round_report, not by copies) */
        if (hipHostMalloc((void**)&c->bs.h_poll, 3 * sizeof(TailPoll), hipHostMallocMapped) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "hipHostMalloc(poll buffer) failed");
        for (int k = 0; k < 2; ++k) if (hipEventCreateWithFlags(&c->bs.poll_ev[k], hipEventDisableTiming) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "hipEventCreate failed");
    }
    if (c->bs.h_jobdyn.size() < 2 * (size_t)nj) c->bs.h_jobdyn.resize(2 * (size_t)nj);
    if (c->bs.h_dyn_cap < 3 * (size_t)nj * sizeof(JobDyn)) {
        if (c->bs.h_dyn) (void)hipHostFree(c->bs.h_dyn);
        c->bs.h_dyn = nullptr; c->bs.h_dyn_cap = 0;
        const size_t want = 3 * (2 * (size_t)nj + 64) * sizeof(JobDyn);      /* twice the views: see DevBuf::reserve */
        if (hipHostMalloc((void**)&c->bs.h_dyn, want, hipHostMallocMapped) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "hipHostMalloc(job poll buffer) failed");
        c->bs.h_dyn_cap = want;
    }
    n_alive = nj;
    mark("setup + uploads (async)");
    return 0;
}

/* ---- round 0: DMRecon::processFeatures (dmrecon.cc:243-331), every SfM feature of every view in one launch */
int BatchRun::seed_round() {
    /* MI_DMRECON_SEED_REOPT (read per call): 2 / unset = the reference's seed semantics, in the seed launch itself; 1 = the same
     * as a round of its own (round 1; same maps, the form the other one is tested against); 0 = every seed propagates at once
     * (the default until round 5: a deviation that can fill pixels the reference's queue never reaches) */
    seed_mode = [] { const char* e = std::getenv("MI_DMRECON_SEED_REOPT"); const int v = e ? std::atoi(e) : 2; return v < 0 || v > 2 ? 2 : v; }();
    if (n_seeds_total == 0) return 0;
    HIP_TRY(hipMemcpyAsync(c->bs.d_work.p, c->bs.h_up + up_seeds, n_seeds_total * sizeof(DevEntry), hipMemcpyHostToDevice, S));
    HIP_TRY(hipMemcpyAsync(c->bs.d_hyp.p, c->bs.h_up + up_hyps, n_seeds_total * sizeof(DevHyp), hipMemcpyHostToDevice, S));
    HIP_TRY(hipMemsetAsync(c->bs.d_keys.p, 0, total_px * sizeof(unsigned long long), S));
    ev.begin(S, EventLog::BULK, (unsigned)n_seeds_total);
    const unsigned ppw = patches_per_wave(st);
    /* The reference's seed semantics (seed_mode 2, the default): every seed that succeeds is optimised once more from its own
     * result in the seed launch itself, and only those whose confidence that strictly raises propagate -- they are stamped as
     * written in round 1, the others as round 0's, k_apply_seeds counts the pixels per view as the size of "round 1", and the
     * propagation starts with round 2: the maps of the two-round form (seed_mode 1: round 1 re-optimises the pixels the seeds
     * wrote), bit for bit, without a round of its own. */
    DevSettings sds = ds;
    sds.seed_reopt = seed_mode == 2 ? 1 : 0;
    D->optimize(S, 1, ((unsigned)n_seeds_total + ppw - 1) / ppw, c->bs.d_jobs.p, c->sc->d_views.p,
                c->sc->d_lut, sds, c->bs.d_work.p, c->bs.d_hyp.p, c->bs.d_results.p, nullptr, (unsigned)n_seeds_total, 0u, 0xFFFFFFFFu, 0,
                c->d_counters, nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr);
    ev.end(S);
    ++n_launch;
    ev.begin(S, EventLog::SWEEP, 0);
    mi_launch_apply_seeds(S, c->bs.d_jobs.p, c->bs.d_work.p, c->bs.d_results.p, (unsigned)n_seeds_total, c->d_counters, c->bs.d_keys.p, c->bs.d_keyoff.p,
                          sds.seed_reopt, c->bs.d_view.p + (size_t)nj);
    ev.end(S);
    if (sds.seed_reopt) round = 2;
    return 0;
}

/* The per-view outcome of a failed / cancelled view (the reference: an exception or a cancel ends THAT DMRecon,
 * apps/dmrecon/dmrecon.cc:314-317, dmrecon.cc:353,101-105): the job is marked dead on the device, its entries are
 * skipped from then on, nothing of it is written back.  Called between rounds / chunks with a read-back of the jobs'
 * flags / n_filled words. */
int BatchRun::poll_views(const JobDyn* dyn, unsigned queue_size) {
    for (int j = 0; j < nj; ++j) {
        const int i = ref_of_job[j];
        if (view_rc[i] != 0) continue;
        int why = 0;
        if ((uint32_t)dyn[j].flags & MI_JOB_EFOOTPRINT) why = MI_DMRECON_EFOOTPRINT;
        else if (progress && progress[i].cancelled) why = MI_DMRECON_ECANCELLED;
        if (progress) { progress[i].filled = dyn[j].n_filled; progress[i].queueSize = queue_size; }
        if (!why) continue;
        view_rc[i] = why; --n_alive;
        if (progress && why == MI_DMRECON_ECANCELLED) progress[i].status = MI_RECON_CANCELLED;
        c->bs.h_jobdyn[2 * j] = (int32_t)((uint32_t)dyn[j].flags | MI_JOB_DEAD);   /* stays valid until the copy has run */
        if (hipMemcpyAsync((char*)(c->bs.d_jobs.p + j) + offsetof(DevJob, flags), &c->bs.h_jobdyn[2 * j], sizeof(int32_t),
                           hipMemcpyHostToDevice, S) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "hipMemcpyAsync(job flags) failed");
    }
    return 0;
}

/* ---- phase A: host-visible rounds, grid sized to the lists: k_generate -> the optimisations -> k_apply.  A round has two
 * lists: the entries of the views that are still in the throughput layout (16 patches per wavefront; first attempts, then
 * the follow-up list) and those of the views that have handed over to the latency layout (k_generate decides per view,
 * from the view's own list sizes: a view's maps do not depend on what it shares a batch with).  Once no view is left in
 * the throughput layout the fused rounds take over (phase B / C: the same lane layout, the same maps bit for bit). */
int BatchRun::bulk_rounds(bool& to_tail) {
    to_tail = false;
    /* MI_DMRECON_ONE_LAUNCH=<entries> (read per call): rounds below this many entries in the throughput layout run as one
     * launch of the general kernel (below) */
    const bool wide = st->nrReconNeighbors > 8;      /* sixteen view slots: the general kernel, one launch per round, nothing speculative */
    const unsigned ONE_LAUNCH_MAX = wide ? 0xFFFFFFFFu : [] { const char* e = std::getenv("MI_DMRECON_ONE_LAUNCH"); return e ? (unsigned)std::max(0L, std::atol(e)) : MI_ONE_LAUNCH_MAX; }();
    const int max_rounds = MI_MAX_ROUNDS - 2 * (int)MI_TAIL_CHUNK - 2;
    unsigned* d_vcount = c->bs.d_view.p; unsigned* d_vmode = d_vcount + 3 * (size_t)nj;
    const unsigned ppw = patches_per_wave(st);
    /* MI_DMRECON_SPEC_ROUNDS=<entries> (read per call; 0 = never): a throughput round that cannot fill the GPU lasts as long
     * as its slowest wavefront -- one whose entry tries two, three, four candidate hypotheses in a row.  Below this size
     * every (entry, rank) pair gets a quad of its own (k_optimize_spec / k_apply_spec: same maps, same counters).  The
     * records are sized for twice the threshold; a round that turns out larger than that runs the plain launches, which
     * are enqueued next to the speculative ones and look at the size on the device. */
    const unsigned SPEC_MAX = wide ? 0u : [] { const char* e = std::getenv("MI_DMRECON_SPEC_ROUNDS"); return e ? (unsigned)std::max(0L, std::atol(e)) : MI_SPEC_ROUNDS; }();
    const unsigned spec_cap = 2u * SPEC_MAX;
    /* MI_DMRECON_SINGLE_FOLLOW=<n> (read per call): 0 = the follow-up list of a large round in ONE launch, all remaining attempts
     * of an entry in a row (round 4's form); n = 1 .. 3: n single-attempt follow-up launches, then one for the rest; 4: every
     * further attempt a launch of its own.  Same maps and counters. */
    const int SINGLE_FOLLOW = [] { const char* e = std::getenv("MI_DMRECON_SINGLE_FOLLOW"); return e ? std::atoi(e) : MI_SINGLE_FOLLOW; }();
    const int FAST_FOLLOW = wide ? 0 : [] { const char* e = std::getenv("MI_DMRECON_FAST_FOLLOW"); return e ? std::max(0, std::min(3, std::atoi(e))) : MI_FAST_FOLLOW; }();
    if (SPEC_MAX > 0 && c->bs.d_spec.reserve(4 * (size_t)spec_cap)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(speculative records) failed");
    /* The rounds are enqueued WITHOUT waiting for their list sizes: every kernel of a round reads its size on the device
     * (k_generate also decides there which layout a view's entries go to), the grids come from the sizes of the last round
     * the host has seen (doubled: a list grows at most ~4x per round in the first rounds, and a grid that is too small
     * only makes its wavefronts stride), and the sizes are read back one round behind -- the host looks at round r - 1
     * while round r runs.  What it decides from them (when to stop, when every view has handed over) may come a round
     * late: the rounds enqueued meanwhile are proper rounds of the same propagation (after the end: empty ones). */
    unsigned known_thr = (unsigned)std::max<size_t>(n_seeds_total, 1) * 4u, known_lat = 0;
    struct Pending { int round; size_t ev_thr, ev_lat; };
    Pending pend[2]; int n_pend = 0;
    bool stop = false;
    /* MI_DMRECON_SEED_REOPT=1 (seed_round): the reference pushes a seed's OWN pixel (dmrecon.cc:316-326) and, when it
     * pops it, re-optimises it from its converged state; only if that strictly raises its confidence is the pixel rewritten
     * and its four neighbours pushed (:365-398).  In this form round 1 is that re-optimisation for every pixel the seeds
     * wrote (the entries are those pixels themselves, their own state the hypothesis, the same strict acceptance), and
     * only the pixels it rewrites are sources of round 2.  (The default does the same inside the seed launch and starts
     * here with round 2: tests/test_gpu_parity.py::test_seed_reoptimisation_round compares the two.) */
    const bool SEED_REOPT = seed_mode == 1;
    auto enqueue = [&](int r) -> int {
        /* a round that will not fill the GPU several times over: speculative launches for lists below spec_cap, the plain
         * ones (below) only above it */
        const bool spec = SPEC_MAX > 0 && known_thr < SPEC_MAX;
        const bool self = SEED_REOPT && r == 1;
        DevSettings ds = this->ds;                                   /* (shadows the member for this round's launches) */
        ds.self_round = self ? 1 : 0;
        ev.begin(S, EventLog::SWEEP, 0);
        D->generate(S, c->bs.d_jobs.p, nj, max_tiles, c->bs.d_work.p, c->bs.d_work2.p, c->bs.d_round_work_t.p, c->bs.d_round_work.p,
                    d_vcount, d_vmode, handover, r, spec ? c->bs.d_follow.p : nullptr, c->bs.d_round_items.p, self ? 1 : 0);
        ev.end(S);
        const unsigned* n_thr_p = c->bs.d_round_work_t.p + r; const unsigned* n_lat_p = c->bs.d_round_work.p + r;
        const unsigned est = std::max(2u * known_thr, 65536u);
        const unsigned waves = (est + ppw - 1) / ppw;
        /* a view can be in the latency layout from round 2 on (from round 1 if told so): which rounds have such entries is
         * decided on the device -- the launches for them are part of every round (empty ones cost microseconds) */
        const bool any_lat = !wide && (r >= 2 || handover >= 1000000000u);
        ev.begin(S, EventLog::BULK, 0);                              /* (entries: filled in at the read-back) */
        const size_t ev_thr = ev.items.size() - 1;
        size_t ev_lat = (size_t)-1;
        unsigned* fcnt = c->bs.d_follow_cnt.p + MI_FOLLOW_LISTS * (size_t)r;
        const unsigned plain_min = spec ? spec_cap : 0u;
        /* a list grows at most four-fold per round (a pixel is in it only if one of its four neighbours was written in the
         * round before), and the host's figure is two rounds old when it enqueues: below a sixteenth of the records' capacity
         * the plain launches cannot be needed */
        const bool need_plain = !spec || 16ull * (unsigned long long)known_thr >= (unsigned long long)spec_cap;
        if (spec) {
            /* (items: ~1.3 per entry; the list of them is the follow-up buffer, which a speculative round does not use) */
            const unsigned quads = std::min(std::max(3u * known_thr, 16384u), 4u * spec_cap);
            D->optimize_spec(S, (quads + ppw - 1) / ppw, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, c->bs.d_spec.p,
                             c->bs.d_follow.p, c->bs.d_round_items.p + r, n_thr_p, 0u, 0u, spec_cap, r, c->d_counters);
        }
        /* Rounds below MI_ONE_LAUNCH_MAX entries run as ONE launch of the general kernel, all attempts of an entry in a
         * row: such a launch fits the GPU at once, so either launch of the two-launch form lasts one wavefront-life
         * (1 attempt, then up to 3 more) whatever its size -- measured: lone 20-view call 533 -> 548 depth-maps/s, a
         * lone 3-view call 19.4 -> 17.8 ms, the bench's plan unchanged (thresholds 50 000 / 200 000 / always: 542 /
         * 546 / 540).  Same arithmetic either way (tests: the maps do not depend on the form); chosen from the last
         * size the host has seen. */
        if (!need_plain) { }
        else if (known_thr < ONE_LAUNCH_MAX)
            D->optimize(S, 1, spec ? std::max(1u, waves / 2) : waves, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr, c->bs.d_results.p, n_thr_p,
                        0u, plain_min, 0xFFFFFFFFu, r, c->d_counters, nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr);
        else {
            /* one optimisation attempt per entry and launch; the entries whose pixel has further candidate hypotheses
             * (about one in five) continue in a follow-up launch over a compacted list (its size stays on the device),
             * so that the wavefronts of both launches are full; the follow-up launch runs all remaining attempts of its
             * entries back to back (third and fourth attempts are rare) */
            /* the follow-up lists: four buffers (the last launch appends to the first one's, consumed by then), each of
             * MI_FOLLOW_SEGS segments -- one per XCD, with a counter of its own (OptArgs::follow_seg) -- of an eighth of the
             * batch's pixels (+ slack: an XCD's eighth of a list is rounded up to whole wavefronts) */
            /* (MI_DMRECON_FOLLOW_SEGMENTS=0, read per call: one list for all XCDs, as until round 6 -- the A/B switch; same maps) */
            const bool segs = [] { const char* e = std::getenv("MI_DMRECON_FOLLOW_SEGMENTS"); return !e || std::atoi(e) != 0; }();
            const unsigned fseg_full = (unsigned)((total_px + MI_FOLLOW_SEGS - 1) / MI_FOLLOW_SEGS) + 64u;
            const unsigned fseg = segs ? fseg_full : 0u;
            const size_t fstride = (size_t)fseg_full * MI_FOLLOW_SEGS;
            unsigned* fl[5] = {c->bs.d_follow.p, c->bs.d_follow.p + fstride, c->bs.d_follow.p + 2 * fstride, c->bs.d_follow.p + 3 * fstride, c->bs.d_follow.p};
            /* The FIRST follow-up list in the order of the round's list (MI_DMRECON_FOLLOW_ORDERED=0: appended by atomics like the
             * later ones -- the A/B switch; same maps): the first launch leaves a mask per wavefront unit, two small kernels make the
             * list from the masks -- a wavefront of it then holds entries of one view and a few image rows instead of 16 strangers,
             * whose footprint gathers share no cache line (1.76 x slower per sampling pass: OptArgs::follow_mask). */
            const bool ordered = SINGLE_FOLLOW && !std::getenv("MI_DMRECON_DEBUG_SCRAMBLE")
                && [] { const char* e = std::getenv("MI_DMRECON_FOLLOW_ORDERED"); return !e || std::atoi(e) != 0; }();
            if (ordered && (c->bs.d_fmask.reserve(total_px / ppw + 4096) || c->bs.d_fblk.reserve(1024)))
                return fail(MI_DMRECON_EDEVICE, "hipMalloc(follow-up masks) failed");
            D->optimize(S, 1, waves, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr, c->bs.d_results.p, n_thr_p,
                        0u, plain_min, 0xFFFFFFFFu, r, c->d_counters, nullptr, nullptr, ordered ? nullptr : fl[0], fcnt, fseg, 0u, ordered ? c->bs.d_fmask.p : nullptr);
            if (ordered) mi_launch_follow_compact(S, c->bs.d_fmask.p, n_thr_p, plain_min, 0xFFFFFFFFu, ppw, 64u / ppw, c->bs.d_fblk.p, fl[0], fcnt);
            if (SINGLE_FOLLOW) {
                /* every further attempt as a launch of its own over the entries the reference's rule still asks one of (about
                 * a fifth, a thirtieth, ... of the list), again ONE attempt per entry: no chain of attempts is live across an
                 * optimisation, hence no spills (the loop form: 304 bytes of scratch per lane), and every wavefront is full.
                 * An entry has at most four candidates, and its first attempt may have been abandoned by the FAST kernel:
                 * four follow-up launches (d_follow holds four lists of a round's entries: BatchScratch::reserve_pixels; the
                 * last launch's own list stays empty) */
                unsigned div = 4;
                /* (MI_DMRECON_SINGLE_FOLLOW=<n>, 1 <= n < 4: n single-attempt follow-up launches, then ONE launch that runs what
                 * is left of its entries' attempts in a row -- a launch lasts a wavefront-life however few entries it has, and the
                 * third and fourth attempts are a few thousandths of the list) */
                const int n_single = SINGLE_FOLLOW >= 4 ? 4 : std::max(1, SINGLE_FOLLOW);
                for (int k = 0; k < n_single; ++k, div *= 4) {
                    /* MI_DMRECON_FAST_FOLLOW=<n>: the first n follow-up launches run the FAST kernel too (second attempts rarely need a
                     * view selection; the ones that do are abandoned again and go on to the next list) */
                    D->optimize(S, k < FAST_FOLLOW ? 2 : 1, std::max(64u, waves / div), c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr,
                                c->bs.d_results.p, n_thr_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters, fl[k], fcnt + MI_FOLLOW_SEGS * k, fl[k + 1], fcnt + MI_FOLLOW_SEGS * (k + 1), fseg,
                                (k == 0 && ordered) ? 0u : fseg, nullptr);
                    ++n_launch;
                }
                if (n_single == 4 && FAST_FOLLOW > 0) {
                    /* an entry abandoned by FAST launches can have attempts left after the four single-attempt launches: what the
                     * last of them appended (to the first list's buffer, consumed long ago) runs its remaining attempts in a row */
                    D->optimize(S, 1, 64u, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr,
                                c->bs.d_results.p, n_thr_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters, fl[4], fcnt + MI_FOLLOW_SEGS * 4, nullptr, nullptr, fseg, fseg, nullptr);
                    ++n_launch;
                }
                if (n_single < 4) {
                    D->optimize(S, 1, std::max(64u, waves / div), c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr,
                                c->bs.d_results.p, n_thr_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters, fl[n_single], fcnt + MI_FOLLOW_SEGS * n_single, nullptr, nullptr, fseg, fseg, nullptr);
                    ++n_launch;
                }
            } else {
                D->optimize(S, 1, std::max(1u, waves / 4), c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work.p, nullptr,
                            c->bs.d_results.p, n_thr_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters, fl[0], fcnt, nullptr, nullptr, fseg, fseg, nullptr);
                ++n_launch;
            }
        }
        ev.end(S);
        ++n_launch;
        if (any_lat) {
            /* the views that have handed over: one wavefront per patch, an entry's attempts one after the other */
            ev.begin(S, EventLog::LAT, 0);
            ev_lat = ev.items.size() - 1;
            /* (the latency list can jump from nothing to a few hundred entries per view in one round, when the views of a batch
             * hand over together: a grid sized from the last list the host has seen would leave a thousand wavefronts striding
             * over it -- measured: 8 ms for such a round in a 200-view batch.  Wavefronts without an entry end at once.) */
            const unsigned lat_grid = std::min(std::max(std::max(4u * known_lat, (unsigned)nj * std::min(handover, 1024u)), 4096u), 32768u);
            D->optimize(S, 16, lat_grid, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, c->bs.d_work2.p,
                        nullptr, c->bs.d_results2.p, n_lat_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters, nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr);
            ev.end(S);
            ++n_launch;
        }
        ev.begin(S, EventLog::SWEEP, 0);
        if (spec) mi_launch_apply_spec(S, std::min((std::max(3u * known_thr, 16384u) + 255) / 256, 768u), c->bs.d_jobs.p, c->bs.d_work.p, c->bs.d_spec.p,
                                       c->bs.d_follow.p, c->bs.d_round_items.p + r, n_thr_p, 0u, 0u, spec_cap, r, c->d_counters);
        if (need_plain) mi_launch_apply(S, std::min((est + 255) / 256, 2048u), c->bs.d_jobs.p, c->bs.d_work.p, c->bs.d_results.p, n_thr_p, 0u, plain_min, 0xFFFFFFFFu, r, c->d_counters);
        if (any_lat) mi_launch_apply(S, std::min((std::max(2u * known_lat, 1024u) + 255) / 256, 4096u), c->bs.d_jobs.p, c->bs.d_work2.p, c->bs.d_results2.p, n_lat_p, 0u, 0u, 0xFFFFFFFFu, r, c->d_counters);
        ev.end(S);
        const int slot = r & 1;
        TailPoll& P = c->bs.h_poll[slot];
        /* the round's report: both list sizes, the counters and the views' flags in one dispatch (k_round_report) */
        mi_launch_round_report(S, n_thr_p, 1, n_lat_p, 1, c->d_counters, c->bs.d_jobs.p, nj, P.rw, &P.hc, dyn_of(slot),
                               d_vcount + (size_t)(r % 3) * nj);
        if (hipGetLastError() != hipSuccess || hipEventRecord(c->bs.poll_ev[slot], S) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "enqueue of a propagation round failed");
        pend[n_pend].round = r; pend[n_pend].ev_thr = ev_thr; pend[n_pend].ev_lat = ev_lat; ++n_pend;
        return 0;
    };
    /* reads back the oldest round in flight; returns 1 when the propagation has ended there */
    unsigned last_thr = 1, last_lat = 0; int last_seen = 0;
    auto retire = [&]() -> int {
        const Pending pd = pend[0];
        pend[0] = pend[1]; --n_pend;
        const int slot = pd.round & 1;
        HIP_TRY(wait_event(c->bs.poll_ev[slot]));
        TailPoll& P = c->bs.h_poll[slot];
        const unsigned n_thr = P.rw[0], n_lat = P.rw[1];
        hc = P.hc;
        if (int rc = poll_views(dyn_of(slot), n_thr + n_lat)) return rc;
        view_list.resize((size_t)nj);
        view_filled.resize((size_t)nj);
        for (int j = 0; j < nj; ++j) { view_list[j] = dyn_of(slot)[j].list; view_filled[j] = dyn_of(slot)[j].n_filled; }
        known_thr = n_thr; known_lat = n_lat;
        last_thr = n_thr; last_lat = n_lat; last_seen = pd.round;
        ev.items[pd.ev_thr].work = n_thr;
        if (pd.ev_lat != (size_t)-1) ev.items[pd.ev_lat].work = n_lat;
        else if (n_lat) return fail(MI_DMRECON_EDEVICE, "internal: latency-layout entries in a round enqueued without their launches");
        if (n_lat) ++n_lat_rounds;
        return (n_thr + n_lat == 0) ? 1 : 0;
    };
    for (; round < max_rounds && n_alive > 0 && !stop; ) {
        if (int rc = enqueue(round)) return rc;
        ++round;
        /* keep ONE round queued behind the one that runs: look at the round before the one just enqueued */
        while (n_pend > 1 || (stop && n_pend > 0)) {
            const int r = retire();
            if (r < 0) return r;
            if (r == 1) {                                          /* an empty round: the propagation is over */
                while (n_pend > 0) { HIP_TRY(wait_event(c->bs.poll_ev[pend[0].round & 1])); pend[0] = pend[1]; --n_pend; }
                round = last_seen; done = true;
                return 0;
            }
            if (n_alive == 0) { HIP_TRY(wait_stream(S)); return 0; }
            /* no view is left in the throughput layout: the fused rounds take over after the rounds already enqueued */
            if (last_thr == 0 && !host_rounds_only) stop = true;
        }
    }
    if (stop) {
        /* every round enqueued has been retired; the last one's latency list is what the fused rounds go on from */
        if (last_thr != 0) return fail(MI_DMRECON_EDEVICE, "internal: throughput entries after the hand-over");
        tail_known = last_lat;
        /* counters as of the end of the host-visible rounds (slot 2 of the poll buffer; read after the call) */
        HIP_TRY(hipMemcpyAsync(&c->bs.h_poll[2].hc, c->d_counters, sizeof(hc), hipMemcpyDeviceToHost, S));
        have_handover = true;
        to_tail = true;
        return 0;
    }
    if (n_alive > 0) truncated = true;               /* round counters exhausted */
    return 0;
}

/* ---- phase B: one fused launch per round (k_tail: candidates from the previous round's accepted entries -> this
 * round's list, optimisations and state writes); lists and results ping-pong.  A chunk of rounds is enqueued blind
 * (the kernels read the round's list size from device memory) and its counters are read back while the NEXT chunk
 * already runs (empty rounds are microsecond no-ops), so the GPU never waits for the host inside the tail.
 * Once the lists are down to a few entries per view the rest goes to the front kernel (phase C). */
int BatchRun::tail_rounds(bool& to_front) {
    to_front = false;
    sparse_r0 = round;            /* every write so far went to the first state slot, with a stamp below this round */
    /* Rounds up to this many entries try a pixel's candidate hypotheses at the same time (four wavefronts per pixel:
     * the round is one patch optimisation long instead of up to four); larger rounds fill the GPU anyway and run
     * them in turn on one wavefront, which wastes nothing.  MI_DMRECON_SPECULATE=<entries> (0 = never). */
    static const unsigned SPEC_MAX = [] { const char* e = std::getenv("MI_DMRECON_SPECULATE"); return e ? (unsigned)std::atoi(e) : 1024u; }();
    const ActiveCall& active = *active_call;
    /* MI_DMRECON_FRONT=<entries per view> (read per call; 0 = never): hand the rest of the propagation to k_front once
     * a round's list is down to that many entries per reference view on average.  There every view runs its rounds
     * at its own pace on 8 x front_team wavefronts; k_tail runs a round of ALL views in one launch as long as its
     * slowest patch -- better while a view's round has many times more entries than its team has wavefronts
     * (measured, lone calls of C3: 20 views, team 12: the whole tail 15.2 ms against 8.4 + 12.4 handed over at 64 per
     * view and 24.4 + 2.3 at 2; 3 views, team 32: 12.2 against 4.8 + 5.8 at 128), hence the default: 64 entries per
     * wavefront-octet of the team, at least 256 -- which for batches of 48+ views (hand-over round: < 12288 entries)
     * is the whole tail.  Next to other calls a front workgroup occupies one CU per view and leaves the rest of the
     * GPU to their bulk rounds, where ~600 launches per batch queue behind them (the bench's plan: 1164-1199
     * depth-maps/s against 980 with one launch per round); a one-view call there must NOT run its whole propagation
     * on one workgroup (the drop-in app's first batches did: 800 ms each). */
    plan_front_team();
    const unsigned FRONT_PER_VIEW = [&] {
        const char* e = std::getenv("MI_DMRECON_FRONT");
        if (e) return (unsigned)std::max(0, std::atoi(e));
        return std::max<unsigned>(MI_FRONT_MIN_CAP, (unsigned)MI_FRONT_PER_TEAM_WG * (unsigned)front_team);
    }();
    const unsigned front_max = FRONT_PER_VIEW * (unsigned)nj;
    wcur = c->bs.d_work2.p; wnext = c->bs.d_work.p; rcur = c->bs.d_results2.p; rnext = c->bs.d_results.p;   /* (the last host-visible round's list) */
    /* the hand-over round's list is already that small: the views go their own ways at once */
    if (front_max > 0 && tail_known <= front_max) { to_front = true; return 0; }
    /* An event record costs ~6 us of queue time on either side of the kernel it brackets -- more than a tenth of a
     * tail round: every 8th round is timed (all of them when tracing); the tail launches are uniform (one dependent
     * patch chain each), their mean stands in for the untimed ones. */
    const unsigned TIMED_EVERY = trace ? 1u : 8u;
    struct ChunkInfo { int first; size_t ev_first, ev_last; };
    ChunkInfo info[2];
    auto enqueue_chunk = [&](int slot) -> int {
        info[slot].first = round; info[slot].ev_first = ev.items.size();
        /* workgroups of a blind round: one per candidate (4 per entry of the previous list) lets the wavefronts without
         * a source end at once; the list sizes are known with a chunk's delay, so the grid is twice what the last
         * known size asks for (a shorter grid strides, correct but with idle wavefronts) */
        const unsigned grid = std::min(65536u, std::max(1024u, 8u * tail_known));
        /* Speculative attempts (workgroups of four wavefronts) buy latency -- for a call that has the GPU to itself or
         * shares it with one other.  Next to the bulk rounds of several other calls the small form (one wavefront per
         * workgroup, 168 registers, 7 KB of LDS) is placed without draining a CU first and takes less from them. */
        const bool speculative = tail_known <= SPEC_MAX && active.count() <= 2;
        for (unsigned k = 0; k < MI_TAIL_CHUNK; ++k, ++round) {
            const bool timed = (stats != nullptr || trace) && k % TIMED_EVERY == 0;
            if (timed) ev.begin(S, EventLog::TAIL, k);            /* k -> entries after the read-back */
            D->tail(S, grid, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, wcur, rcur, wnext, rnext, c->bs.d_round_work.p, round,
                    c->d_counters, speculative);
            if (timed) ev.end(S);
            std::swap(wcur, wnext); std::swap(rcur, rnext);
        }
        info[slot].ev_last = ev.items.size();
        TailPoll& P = c->bs.h_poll[slot];
        mi_launch_round_report(S, c->bs.d_round_work.p + info[slot].first, MI_TAIL_CHUNK, nullptr, 0, c->d_counters, c->bs.d_jobs.p, nj,
                               P.rw, &P.hc, dyn_of(slot), nullptr);
        if (hipGetLastError() != hipSuccess || hipEventRecord(c->bs.poll_ev[slot], S) != hipSuccess)
            return fail(MI_DMRECON_EDEVICE, "enqueue of tail rounds failed");
        return 0;
    };
    int slot = 0;
    if (int rc = enqueue_chunk(0)) return rc;
    for (;;) {
        const bool want_front = front_max > 0 && tail_known <= front_max;
        const bool more_room = round + (int)MI_TAIL_CHUNK < MI_MAX_ROUNDS - 1;
        const bool ahead = more_room && !want_front;              /* keep a second chunk in flight */
        if (ahead) if (int rc = enqueue_chunk(slot ^ 1)) return rc;
        HIP_TRY(wait_event(c->bs.poll_ev[slot]));
        TailPoll& P = c->bs.h_poll[slot];
        hc = P.hc;
        for (size_t q = info[slot].ev_first; q < info[slot].ev_last; ++q) ev.items[q].work = P.rw[ev.items[q].work];
        int end_round = -1;
        unsigned chunk_max = 0;
        for (unsigned k = 0; k < MI_TAIL_CHUNK; ++k) {
            if (P.rw[k] == 0) { end_round = info[slot].first + (int)k; break; }
            ++n_launch; ++n_tail_launch;
            if (k >= MI_TAIL_CHUNK / 2) chunk_max = std::max(chunk_max, P.rw[k]);
        }
        if (chunk_max) tail_known = chunk_max;
        if (int rc = poll_views(dyn_of(slot), P.rw[MI_TAIL_CHUNK - 1])) return rc;
        if (end_round >= 0) {
            /* an empty round: the propagation is over */
            if (ahead) HIP_TRY(wait_event(c->bs.poll_ev[slot ^ 1]));     /* the chunk in flight is all no-ops */
            round = end_round;
            done = true;
            return 0;
        }
        if (n_alive == 0) return 0;
        if (!ahead) {
            /* nothing is in flight behind this chunk: its last round's list is the current one */
            if (want_front) { to_front = true; return 0; }
            truncated = true;                                     /* round counters exhausted */
            return 0;
        }
        slot ^= 1;
    }
}

/* A call that has the GPU to itself gives every view a TEAM of front workgroups (as many as fit the CUs at one workgroup
 * each: all must be resident, they wait for each other every pass).  Not next to other calls of this process (their bulk
 * rounds want the CUs).  What the process cannot see -- another process on the same GPU -- is covered twice: a team
 * launch needs the GPU's TeamToken (an advisory file lock: one team launch per GPU at a time, across processes), and a team
 * whose members do not all show up within MI_DMRECON_TEAM_WAIT_US gives up and the views finish with one workgroup
 * each (front_rounds): slower, never an error.  MI_DMRECON_FRONT_TEAM=<n> (1 = never). */
/* What a front launch with teams looks like (BatchRun::plan_front_team; a pure function of its arguments, so that it can be
 * checked without a GPU: mi_dmrecon_debug_front_teams).
 * A view's team lives on ONE XCD (one L2: k_front): the views are dealt over the XCDs, the CUs of an XCD over its views.
 * When the views do not divide evenly, the XCDs that hold one view fewer have larger teams to give -- 20 views on 8 XCDs:
 * four XCDs with three teams of 10, four with two teams of 16 -- and they go to the views with the most pixels still
 * empty at the hand-over: the call lasts as long as its slowest view, and what is left to fill is what tells the long
 * fronts from the short ones (measured, C3: the four slowest fronts, 11.5-14.4 ms with 10 workgroups each, are among the
 * eight views with the most empty pixels; their lists at the hand-over -- 100 to 170 entries, as everybody's -- say
 * nothing).  Team size does not show in the maps (tests: teams of 1, 2, 3, 8, 25, mixed).
 * want: the largest team allowed; empty: per view the pixels not filled yet, or null (then: by index). */
struct FrontTeams {
    int team_min = 1, team_max = 1;          /* 1 / 1: no teams (some view would be alone anyway) */
    unsigned grid = 0;
    std::vector<unsigned> map;               /* [grid] job | member << 16 | team size << 24, 0xFFFFFFFF = a block without work */
};
FrontTeams build_front_teams(int nj, int n_cus, int want, const long long* empty) {
    FrontTeams ft;
    want = std::min(want, (int)MI_FRONT_TEAM_MAX);
    if (want <= 1 || nj <= 0 || nj >= 65536) return ft;
    const int n_xcd = std::max(1, n_cus / 32), cus_x = std::max(1, n_cus / n_xcd);
    const int base = nj / n_xcd, rem = nj % n_xcd;                  /* XCDs 0 .. rem - 1 hold base + 1 views */
    struct Slot { int xcd, team; };
    std::vector<Slot> slots;
    std::vector<int> team_x((size_t)n_xcd, 0), views_x((size_t)n_xcd, 0);
    int tmin = MI_FRONT_TEAM_MAX, tmax = 1;
    for (int x = 0; x < n_xcd; ++x) {
        views_x[x] = base + (x < rem ? 1 : 0);
        if (views_x[x] == 0) continue;
        team_x[x] = std::max(1, std::min(want, cus_x / views_x[x]));
        tmin = std::min(tmin, team_x[x]);
        tmax = std::max(tmax, team_x[x]);
        for (int k = 0; k < views_x[x]; ++k) slots.push_back(Slot{x, team_x[x]});
    }
    if (tmin <= 1) return ft;
    /* the largest teams to the views with the most empty pixels (ties, and a hand-over the host has no counts of: by index) */
    std::stable_sort(slots.begin(), slots.end(), [](Slot const& a, Slot const& b) { return a.team > b.team; });
    std::vector<int> order((size_t)nj);
    for (int j = 0; j < nj; ++j) order[j] = j;
    if (empty) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return empty[a] > empty[b]; });
    /* block b runs on XCD b % n_xcd; its XCD's blocks in launch order are the members of that XCD's teams, team after team */
    int rows = 0;
    for (int x = 0; x < n_xcd; ++x) rows = std::max(rows, views_x[x] * team_x[x]);
    const unsigned grid = (unsigned)(rows * n_xcd);
    if (grid > 16384u) return ft;
    ft.map.assign(grid, 0xFFFFFFFFu);
    std::vector<int> next_row((size_t)n_xcd, 0);
    for (size_t k = 0; k < slots.size(); ++k) {
        const int j = order[k], x = slots[k].xcd, T = slots[k].team;
        for (int m = 0; m < T; ++m) ft.map[(size_t)(next_row[x] + m) * n_xcd + x] = (unsigned)j | ((unsigned)m << 16) | ((unsigned)T << 24);
        next_row[x] += T;
    }
    ft.grid = grid; ft.team_min = tmin; ft.team_max = tmax;
    return ft;
}

void BatchRun::plan_front_team() {
    const char* e = std::getenv("MI_DMRECON_FRONT_TEAM");
    const int want = e ? std::atoi(e) : (active_call->count() <= 1 ? MI_FRONT_TEAM_MAX : 1);
    std::vector<long long> empty;
    if (view_filled.size() == (size_t)nj) {
        empty.resize((size_t)nj);
        for (int j = 0; j < nj; ++j) empty[j] = (long long)jobs[j].w * jobs[j].h - (long long)view_filled[j];
    }
    FrontTeams ft = build_front_teams(nj, c->n_cus, want, empty.empty() ? nullptr : empty.data());
    front_team = ft.team_min; front_team_max = ft.team_max; front_grid = ft.grid; front_map.swap(ft.map);
}

/* The right to run front teams on a GPU, for as long as the object lives: an exclusive, non-blocking flock on a file
 * named after the GPU's PCI address (the device ordinal differs between processes with different visibility masks).
 * Every attempt opens the file anew, so two calls of one process exclude each other as two processes do.  No file system
 * to put it on, no permission, a file that is not what it should be: no token -- the call runs without teams.
 * The name is predictable and the directory world-writable, so the file is treated as somebody else's: an existing one is
 * opened READ-ONLY (enough for flock) and never followed if it is a symbolic link, it must be a regular file with one
 * link (not a hard link to something of the victim's), and its mode is never touched; only a file this process has just
 * created itself (O_EXCL) gets its mode set -- readable by everybody, so that whoever comes next, another user perhaps,
 * can take the lock.  What a foreign lock holder can do is keep a GPU's calls from running teams: slower, never wrong. */
struct TeamToken {
    int fd = -1;
    static int open_lock(const char* path) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            int f = ::open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC | O_NONBLOCK);
            if (f < 0 && errno == ENOENT) {
                f = ::open(path, O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0644);
                if (f < 0 && errno == EEXIST) continue;              /* somebody else created it meanwhile: open theirs */
                if (f >= 0) (void)::fchmod(f, 0644);                 /* (our own new file: the umask may have taken the read bits) */
            }
            if (f < 0) return -1;
            struct stat sb;
            if (::fstat(f, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_nlink != 1) { ::close(f); return -1; }
            return f;
        }
        return -1;
    }
    explicit TeamToken(int device) {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) std::snprintf(bus, sizeof(bus), "dev%d", device);
        for (char* q = bus; *q; ++q) if (!std::isalnum((unsigned char)*q)) *q = '_';
        const char* dirs[2] = {"/dev/shm", "/tmp"};
        for (int d = 0; d < 2 && fd < 0; ++d) {
            char path[160];
            std::snprintf(path, sizeof(path), "%s/mi_dmrecon_team_%s.lock", dirs[d], bus);
            fd = open_lock(path);
        }
        if (fd >= 0 && ::flock(fd, LOCK_EX | LOCK_NB) != 0) { ::close(fd); fd = -1; }
    }
    ~TeamToken() { if (fd >= 0) { (void)::flock(fd, LOCK_UN); ::close(fd); } }
    bool held() const { return fd >= 0; }
    TeamToken(const TeamToken&) = delete; TeamToken& operator=(const TeamToken&) = delete;
};

/* ---- phase C: the rest of the propagation, one persistent workgroup per reference view (k_front, dmrecon_device.hip):
 * each view runs its own rounds from the list the last tail round left, at its own pace, until its front is empty. */
int BatchRun::front_rounds() {
    unsigned* d_off = c->bs.d_front.p; unsigned* d_cnt = d_off + nj; unsigned* d_stats = d_cnt + nj; unsigned* d_filled = d_stats + 4 * (size_t)nj;
    unsigned long long* d_resume = c->bs.d_front_resume.p;
    HIP_TRY(hipMemcpyAsync(d_off, c->bs.h_up + up_keyoff, nj * sizeof(unsigned), hipMemcpyHostToDevice, S));   /* a view's list region = its pixel offset */
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 6 * (size_t)nj * sizeof(unsigned), S));                          /* sizes, statistics, team counts */
    HIP_TRY(hipMemsetAsync(d_resume, 0, 2 * (size_t)nj * sizeof(unsigned long long), S));
    front_first_round = round;
    const unsigned spin_ticks = [] { const char* e = std::getenv("MI_DMRECON_TEAM_WAIT_US"); return 100u * (e ? (unsigned)std::max(1, std::atoi(e)) : MI_TEAM_WAIT_US); }();
    /* test hook, MI_DMRECON_DEBUG_FRONT_FAULT=<member>[:<round>]: that member of every team vanishes at that round of its view */
    const int fault = [] {
        int f = -1;
        if (const char* e = std::getenv("MI_DMRECON_DEBUG_FRONT_FAULT")) if (*e) {
            const char* colon = std::strchr(e, ':');
            f = (std::atoi(e) & 0xFF) | ((colon ? std::max(0, std::atoi(colon + 1)) & 0xFFFF : 0) << 8);
        }
        /* test hook, MI_DMRECON_DEBUG_TEAM_WT=1: the teams behave as if their members had been found on several XCDs */
        if (const char* e = std::getenv("MI_DMRECON_DEBUG_TEAM_WT")) {
            if (std::atoi(e) == 1) f = (f < 0 ? 0xFF : f) | (1 << 24);
            if (std::atoi(e) == 2) f = (f < 0 ? 0xFF : f) | (1 << 25);       /* 2: exchanges through memory even within one XCD (A/B) */
        }
        return f;
    }();
    std::unique_ptr<TeamToken> token;
    if (front_team > 1) {
        token.reset(new TeamToken(c->device));
        if (!token->held()) { token.reset(); front_team = 1; }     /* another call (process) runs its teams: none for this one */
    }
    if (front_team > 1) {
        if (c->bs.d_front_mail.reserve((size_t)nj * MI_FRONT_MAIL_WORDS) || c->bs.d_front_flags.reserve((size_t)nj * MI_FRONT_FLAG_STRIDE))
            return fail(MI_DMRECON_EDEVICE, "hipMalloc(front mailboxes) failed");
        HIP_TRY(hipMemsetAsync(c->bs.d_front_mail.p, 0, (size_t)nj * MI_FRONT_MAIL_WORDS * sizeof(unsigned long long), S));
        HIP_TRY(hipMemsetAsync(c->bs.d_front_flags.p, 0, (size_t)nj * MI_FRONT_FLAG_STRIDE * sizeof(unsigned), S));
        if (front_map.size() != front_grid || front_grid == 0) return fail(MI_DMRECON_EDEVICE, "internal: front teams without a block map");
        if (c->bs.d_front_map.reserve(front_grid)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(front block map) failed");
        /* (a few hundred words from the call's own memory: the stream is idle here, the copy is staged at once) */
        HIP_TRY(hipMemcpyAsync(c->bs.d_front_map.p, front_map.data(), front_grid * sizeof(unsigned), hipMemcpyHostToDevice, S));
    }
    /* One workgroup per view and more views than the GPU holds front workgroups (one per CU): the launch runs them in waves
     * and lasts as long as its last workgroup -- the views with the most left to fill go first (they are the ones with the
     * long fronts: plan_front_team), the short ones fill the CUs that become free (measured, 400 views: the kernel 56.2 ->
     * 36.6 ms).  What that does not buy in full: the maps of a finished view go back to the host while the kernel still
     * runs (stream_view), 2 MB per view -- 830 MB for 400 views = 31 ms at the 27 GB/s of the box's PCIe link, nearly as long
     * as the front itself --, and longest-first makes the views end together: 12 ms of that transfer then come after the
     * kernel (front + rest of the copies 56.4 -> 48.8 ms).  Measured and dropped: the long quarter first and the others
     * SHORTEST first, so that they end one after the other -- the estimate (empty pixels) does not order the middle of the
     * field well enough, the kernel was 51.5 ms (+ 0.7); two alternating copy streams -- the link is the limit, not the copy
     * engine (profiles/r5_ab_experiments.txt).  MI_DMRECON_FRONT_ORDER=0: index order. */
    const unsigned* d_order = nullptr;
    {
        const char* e = std::getenv("MI_DMRECON_FRONT_ORDER");
        if ((!e || std::atoi(e) != 0) && nj > 1 && view_filled.size() == (size_t)nj) {
            front_order.resize((size_t)nj);
            for (int j = 0; j < nj; ++j) front_order[j] = (unsigned)j;
            auto empty_px = [&](unsigned a) { return (long long)jobs[a].w * jobs[a].h - (long long)view_filled[a]; };
            std::stable_sort(front_order.begin(), front_order.end(), [&](unsigned a, unsigned b) { return empty_px(a) > empty_px(b); });
            if (c->bs.d_front_order.reserve((size_t)nj)) return fail(MI_DMRECON_EDEVICE, "hipMalloc(front order) failed");
            HIP_TRY(hipMemcpyAsync(c->bs.d_front_order.p, front_order.data(), (size_t)nj * sizeof(unsigned), hipMemcpyHostToDevice, S));
            d_order = c->bs.d_front_order.p;
        }
    }
    front_stats.assign(4 * (size_t)nj, 0u);
    /* maps of finished views go back while the others run -- the flags in COHERENT page-locked memory (the default kind is
     * cached on the device: a running kernel's stores to it only show when the kernel ends) -- (not for calls with a progress array: a view that is cancelled
     * after it has ended must not have been written, dmrecon.cc:101-105) */
    unsigned* h_done = nullptr;
    streamed.assign((size_t)nj, 0);
    if (!progress) {
        /* [views] flags | [views][4] the per-view statistics of the kernel (a copy into pageable memory would hold the host
         * in the copy call until the kernel is over) */
        if (c->bs.h_done_cap < 5 * (size_t)nj) {
            if (c->bs.h_done) (void)hipHostFree(c->bs.h_done);
            c->bs.h_done = nullptr; c->bs.h_done_cap = 0;
            if (hipHostMalloc((void**)&c->bs.h_done, (10 * (size_t)nj + 64) * sizeof(unsigned), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) c->bs.h_done_cap = 10 * (size_t)nj + 64;
        }
        if (c->bs.h_done) { h_done = c->bs.h_done; std::memset(h_done, 0, (size_t)nj * sizeof(unsigned)); }
    }
    /* Large batches (MI_DMRECON_SPARSE_MAPS=<views>, default 48; 0 = never): the maps do not wait for their views to end.
     * One workgroup per view and the long views first makes the views end together, and 16 bytes per pixel of every view
     * -- 830 MB for 400 views of C3: 31 ms on the box's PCIe link -- then stood behind the front kernel (16 of 253 ms per
     * batch).  Now: (1) the state as of the hand-over is copied to the caller's buffers WHILE the kernel runs (the first
     * state slot; what the kernel writes under the copy is read torn, and does not matter:) (2) when a view has ended, the
     * pixels written since the hand-over -- either slot, stamp >= sparse_r0: a few per cent -- go to a list in page-locked
     * memory (k_emit_changed), and (3) the host writes them over the snapshot.  Same maps, bit for bit
     * (test_sparse_maps_equal_full_copies).  A list that outgrows its buffer: everything is copied in full afterwards. */
    sparse = false; sparse_overflow = false; emit_order.clear(); emit_seen = 0; sparse_done_end = 0;
    snapped.assign((size_t)nj, 0);
    {
        const char* e = std::getenv("MI_DMRECON_SPARSE_MAPS");
        const int min_views = e ? std::atoi(e) : MI_MERGE_SMALL_CALL;
        if (h_done && min_views > 0 && nj >= min_views && sparse_r0 > 0) {
            bool normal = false;
            for (int i = 0; i < n_refs; ++i) if (maps[i].normal) normal = true;
            sparse_stride = normal ? 9u : 6u;
            BatchScratch& bs = c->bs;
            /* an eighth of the pixels the scratch set holds (C3: the front rewrites 3-4 % of a view's pixels) -- of the SET, not
             * of this batch: the set's buffers have their headroom, and a list sized by the batch was allocated anew (150 MB
             * of page-locked memory: 40 ms) by the first batch larger than the warm-up's */
            const size_t cap = std::min<size_t>(0x7FFFFFFFu, std::max<size_t>(65536, std::max(total_px, bs.pixels()) / 8));
            bool ok = true;
            if (bs.h_sparse_cap < cap * sparse_stride) {
                if (bs.h_sparse) (void)hipHostFree(bs.h_sparse);
                bs.h_sparse = nullptr; bs.h_sparse_cap = 0;
                if (hipHostMalloc((void**)&bs.h_sparse, cap * sparse_stride * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) bs.h_sparse_cap = cap * sparse_stride;
                else { (void)hipGetLastError(); ok = false; }
            }
            if (ok && bs.h_emit_cap < (size_t)nj) {
                if (bs.h_emit_end) (void)hipHostFree(bs.h_emit_end);
                bs.h_emit_end = nullptr; bs.h_emit_cap = 0;
                if (hipHostMalloc((void**)&bs.h_emit_end, 2 * (size_t)nj * sizeof(unsigned), hipHostMallocDefault) == hipSuccess) bs.h_emit_cap = 2 * (size_t)nj;
                else { (void)hipGetLastError(); ok = false; }
            }
            if (ok && bs.d_sparse_count.reserve(1)) ok = false;
            if (ok) {
                sparse_cap = (unsigned)(bs.h_sparse_cap / sparse_stride);
                if (const char* dc = std::getenv("MI_DMRECON_DEBUG_SPARSE_CAP")) sparse_cap = std::min(sparse_cap, (unsigned)std::max(1, std::atoi(dc)));   /* test hook */
                for (int j = 0; j < nj; ++j) bs.h_emit_end[j] = 0xFFFFFFFFu;
                HIP_TRY(hipMemsetAsync(bs.d_sparse_count.p, 0, sizeof(unsigned), S));     /* (before the front kernel, on its stream) */
                sparse = true;
            }
        }
    }
    TailPoll& P = c->bs.h_poll[0];
    const int max_round = round + 4 * MI_MAX_ROUNDS;
    /* the first launch deals the last tail round's list out to the views and runs them (as teams, if any); should the
     * teams give up (a member found no compute unit in time: something else holds them), a second launch takes every
     * unfinished view from the round it stopped in, one workgroup per view */
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool again = attempt == 1;
        ev.begin(S, EventLog::FRONT, tail_known);
        D->front(S, nj, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, ds, wcur, rcur, c->bs.d_round_work.p + (round - 1),
                 wnext, rnext, wcur, rcur, d_off, d_cnt, d_stats, round, max_round, c->d_counters,
                 again ? 1 : front_team, (!again && front_team > 1) ? c->bs.d_front_mail.p : nullptr,
                 (!again && front_team > 1) ? c->bs.d_front_flags.p : nullptr,
                 again ? d_resume : nullptr, again ? d_resume + nj : d_resume, d_filled, spin_ticks, again ? -1 : fault,
                 std::max(1, c->n_cus / 32), h_done, (!again && front_team > 1) ? c->bs.d_front_map.p : nullptr, front_grid, d_order);
        ev.end(S);
        ++n_launch;
        if (h_done) {
            /* while the kernel runs: a view that has ended (it said so in page-locked memory, after writing its state back)
             * is flattened and copied to the caller's buffers on a second stream -- the views end at very different times
             * (C3: between 2.5 and 14.5 ms), only the slowest ones' maps are left when the kernel is over.  (Nothing else is
             * enqueued behind the kernel before this loop: a strided or pageable read-back holds the host in its call.) */
            HIP_TRY(hipEventRecord(c->bs.poll_ev[0], S));
            if (sparse && !again) if (int rc = snapshot_views()) return rc;
            /* test hook, MI_DMRECON_DEBUG_FRONT_EFOOTPRINT=<reference view id>: that view fails DURING the front phase (its footprint
             * flag is raised behind the snapshot copies; the front kernel looks at the flags every few rounds) -- the one way to see
             * what mi_dmrecon.h says about the buffers of a view that ends with an error after the snapshot was taken */
            if (const char* e = std::getenv("MI_DMRECON_DEBUG_FRONT_EFOOTPRINT")) if (*e && !again && sparse) {
                const int want = std::atoi(e);
                for (int j = 0; j < nj; ++j) if (jobs[j].ref_view == want) {
                    c->bs.h_jobdyn[2 * j] = (int32_t)MI_JOB_EFOOTPRINT;      /* (stays valid: the stream is synchronised before the call ends) */
                    HIP_TRY(hipMemcpyAsync((char*)(c->bs.d_jobs.p + j) + offsetof(DevJob, flags), &c->bs.h_jobdyn[2 * j], sizeof(int32_t),
                                           hipMemcpyHostToDevice, c->stream2));
                }
            }
            for (;;) {
                const bool over = hipEventQuery(c->bs.poll_ev[0]) != hipErrorNotReady;
                for (int j = 0; j < nj; ++j)
                    if (!streamed[j] && __atomic_load_n(&h_done[j], __ATOMIC_ACQUIRE) != 0u) {
                        if (int rc = (sparse && snapped[j]) ? emit_view(j) : stream_view(j)) return rc;
                        if (!over) ++n_streamed_early;
                        if (trace) fprintf(stderr, "[mi_dmrecon] view %d streamed back at %.3f ms of the front phase (%s)\n", jobs[j].ref_view, now_ms() - t_mark, over ? "kernel over" : "kernel running");
                    }
                if (sparse) scatter_emitted(false);
                if (over || n_streamed == nj) break;
                /* (a sleep of any length comes back 50+ us later: the timer slack; a small call has nothing better to do
                 * than look again at once, a large batch naps: see PatientWaits) */
                if (g_patient_waits) std::this_thread::sleep_for(std::chrono::microseconds(50));
                else for (int k = 0; k < 200; ++k) cpu_relax();
            }
        }
        if (h_done) {
            mi_launch_round_report(S, d_stats, 4 * nj, nullptr, 0, c->d_counters, c->bs.d_jobs.p, nj, h_done + nj, &P.hc, dyn_of(0), nullptr);
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipMemcpyAsync(&P.hc, c->d_counters, sizeof(hc), hipMemcpyDeviceToHost, S));
            HIP_TRY(read_dyn(0));
            HIP_TRY(hipMemcpyAsync(front_stats.data(), d_stats, 4 * (size_t)nj * sizeof(unsigned), hipMemcpyDeviceToHost, S));
        }
        HIP_TRY(wait_stream(S));
        if (sparse) {
            HIP_TRY(wait_stream(c->stream2));
            scatter_emitted(true);
            if (sparse_overflow)                                  /* the list did not hold it all: full copies after all (download()) */
                for (int j = 0; j < nj; ++j) if (snapped[j] && streamed[j] == 1) { streamed[j] = 0; --n_streamed; snapped[j] = 0; }
            if (trace) fprintf(stderr, "[mi_dmrecon] maps: snapshot of %d views at the hand-over (round %d) + %u changed pixels%s\n",
                               (int)emit_order.size(), sparse_r0, sparse_done_end, sparse_overflow ? " -- list overflow, full copies instead" : "");
        }
        if (h_done) std::memcpy(front_stats.data(), h_done + nj, 4 * (size_t)nj * sizeof(unsigned));
        hc = P.hc;
        token.reset();                                            /* the teams are gone either way */
        if (!(hc.error_flags & 32u)) break;
        if (again) return fail(MI_DMRECON_EDEVICE, "front kernel: inconsistent state after a team gave up (flags %u)", hc.error_flags);
        ++front_fallbacks;
        if (trace) fprintf(stderr, "[mi_dmrecon] front teams of %d gave up (a member did not show up within %u us): finishing with one workgroup per view\n",
                           front_team, spin_ticks / 100u);
        hc.error_flags &= ~32u;
        c->bs.h_jobdyn.resize(std::max<size_t>(c->bs.h_jobdyn.size(), 2 * (size_t)nj + 1));
        c->bs.h_jobdyn[2 * (size_t)nj] = (int32_t)hc.error_flags;   /* stays valid until the copy has run (synchronised below) */
        HIP_TRY(hipMemcpyAsync((char*)c->d_counters + offsetof(DevCounters, error_flags), &c->bs.h_jobdyn[2 * (size_t)nj], sizeof(unsigned), hipMemcpyHostToDevice, S));
    }
    ran_front = true;
    /* n_rounds as one launch per round counts them: up to the first round that accepted nothing (the slowest view's
     * last round is that one) */
    unsigned rmax = 0;
    for (int j = 0; j < nj; ++j) rmax = std::max(rmax, front_stats[4 * j]);
    if (rmax > 0) round += (int)rmax - 1;
    if (int rc = poll_views(dyn_of(0), 0)) return rc;
    if (hc.error_flags & 8u) truncated = true;
    else done = true;
    return 0;
}

/* The maps as of the hand-over, every view of the batch, to the callers' buffers on the second stream while the front
 * kernel runs on the first (front_rounds).  The first state slot holds all of it: the host-visible rounds only write that. */
int BatchRun::snapshot_views() {
    hipStream_t S2 = c->stream2;
    for (int j = 0; j < nj; ++j) {
        const int i = ref_of_job[j];
        if (view_rc[i] != 0 || streamed[j]) continue;
        mi_dmrecon_maps& m = maps[i];
        if (m.views) continue;                                /* (the local view sets: download(), as for a streamed view) */
        const size_t np = (size_t)jobs[j].w * jobs[j].h;
        if (m.depth) HIP_TRY(hipMemcpyAsync(m.depth, dj[j].depth, np * 4, hipMemcpyDeviceToHost, S2));
        if (m.conf) HIP_TRY(hipMemcpyAsync(m.conf, dj[j].conf, np * 4, hipMemcpyDeviceToHost, S2));
        if (m.dz) HIP_TRY(hipMemcpyAsync(m.dz, dj[j].dz, np * 8, hipMemcpyDeviceToHost, S2));
        if (m.normal) HIP_TRY(hipMemcpyAsync(m.normal, dj[j].normal, np * 12, hipMemcpyDeviceToHost, S2));
        snapped[j] = 1;
    }
    return 0;
}

/* A view that has ended: the pixels written since the snapshot, appended to the list (second stream, behind the snapshot
 * copies), and the list's length behind them. */
int BatchRun::emit_view(int j) {
    streamed[j] = 1; ++n_streamed;
    hipStream_t S2 = c->stream2;
    mi_launch_emit_changed(S2, c->bs.d_maps.p, c->bs.d_imaps.p, total_px, jobs[j].pix_off, (size_t)jobs[j].w * jobs[j].h, sparse_r0,
                           (unsigned)j, sparse_stride, c->bs.d_sparse_count.p, sparse_cap, c->bs.h_sparse);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&c->bs.h_emit_end[emit_order.size()], c->bs.d_sparse_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, S2));
    emit_order.push_back(j);
    return 0;
}

/* Writes the records that have arrived over the snapshot (a view's snapshot copies are ahead of its records on the second
 * stream: they have landed).  all = false: only when enough has gathered to be worth a team of threads. */
void BatchRun::scatter_emitted(bool all) {
    unsigned end = sparse_done_end;
    size_t seen = emit_seen;
    while (seen < emit_order.size()) {
        const unsigned e = __atomic_load_n(&c->bs.h_emit_end[seen], __ATOMIC_ACQUIRE);
        if (e == 0xFFFFFFFFu) break;
        end = e; ++seen;
    }
    if (seen == emit_seen) return;
    if (!all && seen - emit_seen < 32 && end - sparse_done_end < (1u << 18)) return;
    emit_seen = seen;
    if (end > sparse_cap) { sparse_overflow = true; end = sparse_cap; }
    const unsigned first = sparse_done_end;
    if (end <= first) return;
    sparse_done_end = end;
    const uint32_t* rec = c->bs.h_sparse;
    const unsigned stride = sparse_stride;
    const long long n = (long long)end - (long long)first;
    const int n_threads = (int)std::max<long long>(1, std::min<long long>(host_threads_cap(), n / 8192));
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
    for (long long k = 0; k < n; ++k) {
        const uint32_t* r = rec + (size_t)(first + k) * stride;
        const unsigned j = r[0];
        const size_t p = r[1];
        if (j >= (unsigned)nj) continue;
        const int i = ref_of_job[j];
        if (view_rc[i] != 0 || p >= (size_t)jobs[j].w * jobs[j].h) continue;
        mi_dmrecon_maps& m = maps[i];
        if (m.depth) std::memcpy(&m.depth[p], &r[2], 4);
        if (m.conf) std::memcpy(&m.conf[p], &r[3], 4);
        if (m.dz) std::memcpy(&m.dz[2 * p], &r[4], 8);
        if (stride >= 9 && m.normal) std::memcpy(&m.normal[3 * p], &r[6], 12);
    }
}

/* One finished view's maps to the caller's buffers on the second stream (the front kernel still runs on the first): its
 * second state slot folded in, then the copies.  The local view sets (host post-processing) stay for download(). */
int BatchRun::stream_view(int j) {
    streamed[j] = 1; ++n_streamed;
    const int i = ref_of_job[j];
    if (view_rc[i] != 0) return 0;
    mi_dmrecon_maps& m = maps[i];
    if (m.views) { streamed[j] = 2; return 0; }              /* (2: ended, but everything is left to download()) */
    const size_t np = (size_t)jobs[j].w * jobs[j].h;
    hipStream_t S2 = c->stream2;
    mi_launch_flatten(S2, c->bs.d_maps.p, c->bs.d_imaps.p, total_px, st->nrReconNeighbors > 4, jobs[j].pix_off, np);
    if (m.depth) HIP_TRY(hipMemcpyAsync(m.depth, dj[j].depth, np * 4, hipMemcpyDeviceToHost, S2));
    if (m.conf) HIP_TRY(hipMemcpyAsync(m.conf, dj[j].conf, np * 4, hipMemcpyDeviceToHost, S2));
    if (m.dz) HIP_TRY(hipMemcpyAsync(m.dz, dj[j].dz, np * 8, hipMemcpyDeviceToHost, S2));
    if (m.normal) HIP_TRY(hipMemcpyAsync(m.normal, dj[j].normal, np * 12, hipMemcpyDeviceToHost, S2));
    return 0;
}

/* ---- results back to the caller's buffers (views that did not finish keep their buffers untouched) */
int BatchRun::download() {
    std::vector<uint32_t> packed;
    for (int i = 0; i < n_refs; ++i) {
        const int j = job_of[i];
        if (j < 0 || view_rc[i] != 0) continue;
        if (!streamed.empty() && streamed[j] == 1) continue;      /* went back while the front kernel ran */
        if (progress) progress[i].status = MI_RECON_SAVING;
        const size_t np = (size_t)jobs[j].w * jobs[j].h;
        mi_dmrecon_maps& m = maps[i];
        if (m.depth) HIP_TRY(hipMemcpyAsync(m.depth, dj[j].depth, np * 4, hipMemcpyDeviceToHost, S));
        if (m.conf) HIP_TRY(hipMemcpyAsync(m.conf, dj[j].conf, np * 4, hipMemcpyDeviceToHost, S));
        if (m.dz) HIP_TRY(hipMemcpyAsync(m.dz, dj[j].dz, np * 8, hipMemcpyDeviceToHost, S));
        if (m.normal) HIP_TRY(hipMemcpyAsync(m.normal, dj[j].normal, np * 12, hipMemcpyDeviceToHost, S));
        if (m.views) {
            const int nch = mi_dmrecon_local_view_channels(st->nrReconNeighbors);
            for (int half = 0; half < std::min(nch, 8) / 4; ++half) {
                packed.resize(np);
                HIP_TRY(hipMemcpyAsync(packed.data(), half ? dj[j].views_hi : dj[j].views, np * 4, hipMemcpyDeviceToHost, S));
                HIP_TRY(wait_stream(S));
                for (size_t p = 0; p < np; ++p)
                    for (int k = 0; k < 4; ++k) {
                        const unsigned g = (packed[p] >> (8 * k)) & 0xFFu;
                        m.views[(size_t)nch * p + 4 * half + k] = (g == MI_VIEW_NONE || g >= jobs[j].global.size()) ? -1 : jobs[j].global[g];
                    }
            }
            if (nch == 16) {                                      /* view slots 8..15: two words per pixel (DevJob::views_x) */
                packed.resize(2 * np);
                HIP_TRY(hipMemcpyAsync(packed.data(), dj[j].views_x, np * 8, hipMemcpyDeviceToHost, S));
                HIP_TRY(wait_stream(S));
                for (size_t p = 0; p < np; ++p)
                    for (int k = 0; k < 8; ++k) {
                        const unsigned g = (packed[2 * p + (k >> 2)] >> (8 * (k & 3))) & 0xFFu;
                        m.views[(size_t)nch * p + 8 + k] = (g == MI_VIEW_NONE || g >= jobs[j].global.size()) ? -1 : jobs[j].global[g];
                    }
            }
        }
    }
    HIP_TRY(wait_stream(S));
    if (n_streamed) HIP_TRY(wait_stream(c->stream2));
    mark("download");
    return 0;
}

void BatchRun::fill_stats() {
    if (stats) {
        stats->n_patch = (int64_t)hc.n_patch; stats->n_eval = (int64_t)hc.n_eval; stats->n_pass = (int64_t)hc.n_pass;
        stats->n_filled = (int64_t)hc.n_filled;
        stats->n_seeds = (int64_t)n_seed_feats; stats->n_seeds_ok = (int64_t)hc.n_seeds_ok;
        stats->n_rounds = round; stats->n_launches = n_launch; stats->truncated = truncated ? 1 : 0;
        stats->n_view_replaced = (int64_t)hc.n_view_replaced; stats->n_iter14 = (int64_t)hc.n_iter14;
        for (int k = 0; k < 8; ++k) {
            stats->n_eval_by_kernel[k] = (int64_t)hc.k_eval[k]; stats->n_pass_by_kernel[k] = (int64_t)hc.k_pass[k];
            stats->n_patch_by_kernel[k] = (int64_t)hc.k_patch[k]; stats->n_pass_executed_by_kernel[k] = (int64_t)hc.k_pass_exec[k];
        }
        stats->clk_shader_cycles = (int64_t)hc.clk_shader; stats->clk_real_ticks = (int64_t)hc.clk_real;
        stats->clk_real_mhz = c->wall_clock_khz * 1e-3;
        stats->shader_clock_mhz = hc.clk_real ? (double)hc.clk_shader / (double)hc.clk_real * stats->clk_real_mhz : 0.0;
        double tail_ms = 0.0; int64_t tail_timed = 0;
        for (const EventLog::Item& it : ev.items) {
            float ms = 0.f;
            if (!ev.ms(it, ms)) continue;
            switch (it.kind) {
                case EventLog::SWEEP: stats->ms_sweep_kernels += ms; break;
                case EventLog::BULK: stats->ms_bulk_kernel += ms; if (it.work) ++stats->n_bulk_launches; break;   /* (empty launches of blind rounds: their microseconds count, they do not) */
                case EventLog::LAT: stats->ms_bulk_kernel += ms; stats->ms_latency_rounds += ms; stats->n_latency_entries += it.work; if (it.work) ++stats->n_bulk_launches; break;
                case EventLog::TAIL: if (it.work > 0) { tail_ms += ms; ++tail_timed; } break;
                case EventLog::FRONT: stats->ms_front_kernel += ms; break;
            }
        }
        /* phase B: mean of the timed launches x number of launches that had work */
        if (tail_timed > 0) stats->ms_tail_kernel = tail_ms / (double)tail_timed * (double)n_tail_launch;
        stats->ms_opt_kernel = stats->ms_bulk_kernel + stats->ms_tail_kernel + stats->ms_front_kernel;
        stats->n_tail_launches = n_tail_launch;
        stats->n_latency_rounds = n_lat_rounds;
        stats->n_patch_turns = (int64_t)hc.n_stage; stats->n_wave_turns = (int64_t)hc.n_gather_pass;
        if (ran_front) {
            stats->n_front_launches = 1;
            stats->front_team = front_team;
            stats->front_team_max = front_team > 1 ? front_team_max : 1;
            stats->n_sparse_records = sparse ? (sparse_overflow ? -1 : (int64_t)sparse_done_end) : 0;
            stats->front_fallbacks = front_fallbacks;
            stats->front_first_round = front_first_round;
            for (int j = 0; j < nj; ++j) {
                const unsigned* fs = &front_stats[4 * (size_t)j];
                if (fs[0]) ++stats->n_front_views;
                stats->n_front_rounds_max = std::max<int64_t>(stats->n_front_rounds_max, fs[0]);
                stats->n_front_rounds_sum += fs[0]; stats->n_front_attempts += fs[1]; stats->n_front_entries += fs[2];
                stats->ms_front_view_max = std::max(stats->ms_front_view_max, fs[3] * 1e-5);   /* 100 MHz ticks */
            }
        }
        const DevCounters& ho = have_handover ? c->bs.h_poll[2].hc : hc;      /* the stream has been synchronised */
        stats->n_eval_bulk = (int64_t)ho.n_eval; stats->n_patch_bulk = (int64_t)ho.n_patch; stats->n_filled_bulk = (int64_t)ho.n_filled;
        stats->ms_total = now_ms() - t_begin;
    }
    if (trace) {
        size_t w = 0;
        for (const EventLog::Item& it : ev.items) {
            if (it.kind == EventLog::SWEEP) continue;
            float ms = 0.f;
            (void)ev.ms(it, ms);
            if (it.work || ms > 0.02f)
                fprintf(stderr, "[mi_dmrecon] %s launch %zu: %u entries, %.3f ms\n", it.kind == EventLog::FRONT ? "front" : "optimise", w, it.work, ms);
            ++w;
        }
        if (ran_front)
            for (int j = 0; j < nj; ++j) {
                const unsigned* fs = &front_stats[4 * (size_t)j];
                unsigned team = 1;
                for (size_t b = 0; b < front_map.size() && front_team > 1; ++b)
                    if (front_map[b] != 0xFFFFFFFFu && (int)(front_map[b] & 0xFFFFu) == j) { team = front_map[b] >> 24; break; }
                fprintf(stderr, "[mi_dmrecon] front view %d: %u rounds, %u attempts, %u entries, %.3f ms, team of %u, list at the hand-over %u, filled %u of %d\n", jobs[j].ref_view,
                        fs[0], fs[1], fs[2], fs[3] * 1e-5, team, view_list.size() == (size_t)nj ? view_list[j] : 0u,
                        view_filled.size() == (size_t)nj ? view_filled[j] : 0u, jobs[j].w * jobs[j].h);
            }
        fprintf(stderr, "[mi_dmrecon] total %.2f ms host wall\n", now_ms() - t_begin);
    }
}

/* per-view statuses and the call's return value */
int BatchRun::outcome() {
    int n_ok = 0, first_rc = 0, first_i = -1;
    for (int i = 0; i < n_refs; ++i) {
        if (status_out) status_out[i] = view_rc[i];
        if (view_rc[i] == 0) { ++n_ok; if (progress) progress[i].status = MI_RECON_IDLE; }
        else if (!first_rc || (first_rc == MI_DMRECON_ECANCELLED && view_rc[i] != MI_DMRECON_ECANCELLED)) { first_rc = view_rc[i]; first_i = i; }
    }
    if (truncated) return fail(MI_DMRECON_EDEVICE, "propagation did not finish within %d rounds", MI_MAX_ROUNDS);
    /* device-side diagnostics other than the per-view ones (bit 0: a footprint exception, reported per view; bit 3: the
     * truncation above): none of them can happen -- if one did, the maps are not to be trusted */
    if (hc.error_flags & ~(1u | 8u)) return fail(MI_DMRECON_EDEVICE, "device error flags %u (bit 4: a stale mailbox read in a front team)", hc.error_flags);
    if (n_ok > 0) return 0;
    /* every view of the call failed: the first failure's own code and message (a single-view call behaves like
     * DMRecon::start(): the exception / the cancellation is the call's outcome) */
    switch (first_rc) {
        case MI_DMRECON_ECANCELLED: return fail(first_rc, "cancelled");
        case MI_DMRECON_EFOOTPRINT: return fail(first_rc, "Negative pixel footprint");
        case MI_DMRECON_EGVS: return fail(first_rc, "Global View Selection failed");
        case MI_DMRECON_ENOIMAGE:
            if (first_i >= 0 && !plan_err[first_i].empty()) { g_err = plan_err[first_i]; return first_rc; }
            return fail(first_rc, "a selected neighbour view has no image");
        default:
            if (first_i >= 0 && !plan_err[first_i].empty()) { g_err = plan_err[first_i]; return first_rc; }
            return fail(first_rc ? first_rc : MI_DMRECON_EDEVICE, "reconstruction failed");
    }
}

}  // namespace

/* A call's lease on a scratch set of its scene (best fit), back to the pool when the call ends.  A set is only ever
 * created when every existing one is in use, so a scene owns as many as it has had calls in flight at once. */
struct ScratchLease {
    mi_dmrecon_ctx* c;
    ScratchLease(mi_dmrecon_ctx* c_, size_t pixels) : c(c_) {
        std::lock_guard<std::mutex> lock(c->sc->pool_mu);
        std::vector<BatchScratch>& pool = c->sc->scratch_pool;
        if (c->bs.holds_anything()) pool.push_back(std::move(c->bs));   /* (the parity hooks allocate without a lease) */
        c->bs = BatchScratch();
        if (pool.empty()) return;                         /* the first call of the scene, or every set is in use: a new one */
        /* the smallest free set that holds the batch (the large ones stay for the large batches); if none does, the
         * largest: growing it is cheaper than growing a small one */
        size_t pick = 0;
        for (size_t i = 1; i < pool.size(); ++i) {
            const size_t a = pool[i].pixels(), b = pool[pick].pixels();
            if (b >= pixels ? (a >= pixels && a < b) : a > b) pick = i;
        }
        c->bs = std::move(pool[pick]);
        pool.erase(pool.begin() + (std::ptrdiff_t)pick);
    }
    ~ScratchLease() {
        /* nothing of this call may still be queued when the set goes back (the early returns of a failed call leave
         * kernels and copies in flight; the next holder would write into them, or free them) */
        (void)wait_stream(c->stream);
        if (c->stream2) (void)wait_stream(c->stream2);
            std::lock_guard<std::mutex> lock(c->sc->pool_mu);
        if (c->bs.holds_anything()) c->sc->scratch_pool.push_back(std::move(c->bs));
        c->bs = BatchScratch();
    }
};

static int reconstruct_batch(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t n_refs, const int32_t* ref_views,
                             mi_dmrecon_maps* maps, mi_dmrecon_progress* progress, int32_t* status_out,
                             mi_dmrecon_stats* stats) {
    if (!c || !ref_views || !maps || n_refs <= 0) return fail(MI_DMRECON_EINVAL, "null argument");
    int rc = check_settings(st);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const ActiveCall active_call(c->device);
    if (stats) std::memset(stats, 0, sizeof(*stats));
    /* pixels of the batch, for the choice of the scratch set (an estimate is enough: the buffers grow if it was low) */
    size_t px_hint = 0;
    for (int32_t i = 0; i < n_refs; ++i) {
        const int32_t v = ref_views[i];
        if (v < 0 || (size_t)v >= c->sc->views.size()) continue;
        const HostView& hv = c->sc->views[v];
        const size_t l = std::min<size_t>((size_t)std::max<int32_t>(0, st->scale), hv.levels.empty() ? 0 : hv.levels.size() - 1);
        if (!hv.levels.empty()) px_hint += (size_t)hv.levels[l].w * (size_t)hv.levels[l].h;
    }
    ScratchLease lease(c, px_hint);
    PatientWaits patient(n_refs >= MI_MERGE_SMALL_CALL);
    BatchRun B;
    B.c = c; B.st = st; B.n_refs = n_refs; B.ref_views = ref_views; B.maps = maps; B.progress = progress;
    B.status_out = status_out; B.stats = stats; B.D = mi_device_api(st->filterWidth); B.ds = dev_settings(st); B.S = c->stream;
    B.trace = std::getenv("MI_DMRECON_TRACE") != nullptr; B.t_begin = B.t_mark = now_ms();
    B.ev.c = c; B.active_call = &active_call;
    std::memset(&B.hc, 0, sizeof(B.hc));
    try {
        if ((rc = B.plan()) != 0) return rc;
        double t_ph = now_ms();
        auto phase = [&](double mi_dmrecon_stats::*field) { const double t = now_ms(); if (stats) stats->*field += t - t_ph; t_ph = t; };
        if ((rc = B.upload()) != 0) return rc;
        phase(&mi_dmrecon_stats::ms_wall_setup);
        if ((rc = B.seed_round()) != 0) return rc;
        for (int i = 0; progress && i < n_refs; ++i) if (B.view_rc[i] == 0) progress[i].status = MI_RECON_QUEUE;
        /* the propagation sweeps (replace DMRecon::processQueue, dmrecon.cc:333-434) */
        bool to_tail = false, to_front = false;
        if ((rc = B.bulk_rounds(to_tail)) != 0) return rc;
        B.mark("seeds + phase A rounds");
        if (to_tail) {
            if ((rc = B.tail_rounds(to_front)) != 0) return rc;
            B.mark("phase B rounds");
            phase(&mi_dmrecon_stats::ms_wall_rounds);
            if (to_front) { if ((rc = B.front_rounds()) != 0) return rc; B.mark("phase C (front kernel)"); phase(&mi_dmrecon_stats::ms_wall_front); }
            mi_launch_flatten(B.S, c->bs.d_maps.p, c->bs.d_imaps.p, B.total_px, st->nrReconNeighbors > 4, 0, B.total_px);
        }
        if ((rc = B.download()) != 0) return rc;
        phase(&mi_dmrecon_stats::ms_wall_download);
        B.fill_stats();
        return B.outcome();
    } catch (const std::bad_alloc&) {
        return fail(MI_DMRECON_EDEVICE, "out of host memory");
    }
}

/*
 * The entry point.  Calls that arrive at the same time on contexts of one scene (several host threads, each with a
 * forked context) and ask for the same settings are MERGED into one batch: the reference views of a batch are
 * independent jobs, so every caller gets the maps of its own views (up to what any change of batch composition does:
 * the round at which a batch changes lane layouts, i.e. rounding -- tests/test_gpu_fullsize.py), but the batch pays the
 * ~600 latency-bound tail rounds once and its bulk launches are large (one thread with 400 views per call reaches
 * 905 depth-maps/s where six threads with 100 each reach 740-830).  This is what the shim does
 * for mvs::DMRecon::start(); here for callers of the C ABI.
 *   - A call that finds fewer than MI_DMRECON_MERGE_RUNNING (2) batches running becomes a LEADER: it takes every
 *     request pending at that moment (after a wait of MI_DMRECON_MERGE_WINDOW_US = 1000 / 3000 us for more, only if calls of
 *     this scene have met within the last few calls), runs them as one batch on its own context and hands the results out.  The other
 *     calls wait; whoever is still pending when a batch ends becomes the next leader.  A lone caller never waits.
 *   - Per-view statuses go to their callers; a caller all of whose views failed gets the first failure as its return
 *     code, as from its own call.  The statistics go to the leader (n_merged_calls = calls served); the other
 *     callers get zeros with merged_into_other_call = 1, so that sums over calls stay right.
 *   - Calls with a progress array (cancellation, status polling) are never merged.  MI_DMRECON_MERGE_CALLS=0: off.
 */
static int mi_dmrecon_reconstruct_impl(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t n_refs, const int32_t* ref_views,
                           mi_dmrecon_maps* maps, mi_dmrecon_progress* progress, int32_t* status_out,
                           mi_dmrecon_stats* stats) {
    /* (read per call: tests switch them) */
    const bool MERGE = [] { const char* e = std::getenv("MI_DMRECON_MERGE_CALLS"); return e ? std::atoi(e) != 0 : true; }();
    /* Batches of one scene in flight at a time.  Two, measured: their host-visible rounds interleave (a launch of one
     * fills the drain of the other's and its host round trip) -- taking turns in phase A instead costs 15 % at the
     * bench's plan, four in flight 30 % */
    const int MAX_RUNNING = [] { const char* e = std::getenv("MI_DMRECON_MERGE_RUNNING"); return e ? std::max(1, std::atoi(e)) : 2; }();
    /* MI_DMRECON_MERGE_SPLIT=<g> (read per call; experiment): a leader takes at most 1/g of the calls that are there */
    const int SPLIT = [] { const char* e = std::getenv("MI_DMRECON_MERGE_SPLIT"); return e ? std::max(1, std::atoi(e)) : 1; }();
    const int WINDOW_ENV = [] { const char* e = std::getenv("MI_DMRECON_MERGE_WINDOW_US"); return e ? std::max(0, std::atoi(e)) : -1; }();
    /* A call gains from company: the launches of a larger batch fill the GPU better, the latency-bound tail is paid once
     * per batch, and ONE batch of all the callers' views beats two batches side by side (round 4, the bench's plan of four
     * 100-view calls: one batch of 400 views 1 335-1 410 depth-maps/s, 1 + 3 calls 1 135-1 300, 1 + 1 + 2 calls 1 100-1 270).
     * A leader that expects company therefore waits for it -- a millisecond if its own call is small (< 48 views: 3-4 % of
     * such a call), three if it is large (~2 %); that also catches the callers that come back from the batch that has just
     * ended.  A caller that has been alone lately does not wait at all (company_credit). */
    const int WINDOW_US = WINDOW_ENV >= 0 ? WINDOW_ENV : (n_refs < MI_MERGE_SMALL_CALL ? MI_MERGE_WINDOW_US : MI_MERGE_WINDOW_BIG_US);
    {
        /* The host planning of a batch is an OpenMP loop on the thread that LEADS the batch, and a thread's OpenMP team is
         * created at its first parallel region (10-30 ms for 64 threads): which caller leads a merged batch is a matter of
         * arrival order, so every calling thread gets its team at its first call, whatever its role in it -- a team of
         * the size the planning of THIS call would use (the views of the call, the cores the process may run on: not 64
         * idle workers per caller thread on a container with the CPU time of 16). */
        static thread_local bool team_ready = false;
        if (!team_ready) {
            team_ready = true;
            const int n_threads = std::max(1, std::min(std::max(n_refs, 1) * 4, host_threads_cap()));
#pragma omp parallel num_threads(n_threads)
            { }
        }
    }
    if (!MERGE || progress || !c || !st || !ref_views || !maps || n_refs <= 0)
        return reconstruct_batch(c, st, n_refs, ref_views, maps, progress, status_out, stats);
    MergeQueue& Q = c->sc->merge;
    MergeReq me;
    me.st = st; me.n = n_refs; me.refs = ref_views; me.maps = maps; me.status = status_out; me.stats = stats;
    std::vector<MergeReq*> batch;
    {
        std::unique_lock<std::mutex> lock(Q.mu);
        Q.pending.push_back(&me);
        if (Q.gathering) Q.cv.notify_all();                                 /* (the leader counts who is there) */
        if (Q.pending.size() > 1 || Q.running > 0) Q.company_credit = 8;   /* company now: expect it for the next few calls */
        else if (Q.company_credit > 0) --Q.company_credit;                  /* a caller that stays alone stops waiting */
        /* wait until my request has been served by another leader, or I can lead */
        Q.cv.wait(lock, [&] { return me.done || (!me.taken && Q.running < MAX_RUNNING && !Q.gathering); });
        if (me.done) { if (me.rc) g_err = me.err; return me.rc; }
        ++Q.running;
        me.taken = true;
        Q.pending.erase(std::find(Q.pending.begin(), Q.pending.end(), &me));
        batch.push_back(&me);
        if (Q.company_credit > 0 && WINDOW_US > 0) {              /* others are probably on their way: let them join me */
            Q.gathering = true;                              /* (nobody else starts to lead meanwhile) */
            /* ... until the window has run out, or every call that can come is there (MergeQueue::expect; the four callers of
             * the bench's plan arrive within a few hundred microseconds of each other: the rest of the 3 ms was 1 % of a batch) */
            /* (test hook MI_DMRECON_DEBUG_MERGE_WAIT_FOR=<calls>: the leader gathers until that many calls -- its own included, of
             * any settings -- are there, or the window runs out: a deterministic batch for tests/test_gpu_parity.py) */
            const int WAIT_FOR = [] { const char* e = std::getenv("MI_DMRECON_DEBUG_MERGE_WAIT_FOR"); return e ? std::atoi(e) : 0; }();
            (void)Q.cv.wait_for(lock, std::chrono::microseconds(WINDOW_US),
                                [&] { return WAIT_FOR > 0 ? (int)Q.pending.size() + 1 >= WAIT_FOR
                                                          : (Q.expect > 1 && (int)Q.pending.size() + 1 >= Q.expect - Q.running_calls); });
            Q.gathering = false;
        }
        /* take every pending request with my settings, in arrival order.  Measured and dropped: taking only half of them
         * next to a running batch (two half-size batches side by side afterwards: 970-1 040 against 1 095-1 210 at the
         * bench's plan), and taking half of them with both slots free (2 + 2 calls instead of 1 + 1 + 2: 1 050-1 290,
         * mean below 1 + 1 + 2) */
        const size_t take_max = SPLIT > 1 ? (Q.pending.size() + 1 + (size_t)SPLIT - 1) / (size_t)SPLIT : (size_t)-1;
        for (size_t i = 0; i < Q.pending.size() && batch.size() < take_max;) {
            MergeReq* r = Q.pending[i];
            if (std::memcmp(r->st, st, sizeof(*st)) == 0) { r->taken = true; batch.push_back(r); Q.pending.erase(Q.pending.begin() + i); }
            else ++i;
        }
        Q.recent[Q.recent_i++ & 7] = (int)batch.size() + (int)Q.pending.size() + Q.running_calls;
        Q.expect = *std::max_element(Q.recent, Q.recent + 8);
        Q.running_calls += (int)batch.size();
    }
    Q.cv.notify_all();                                       /* requests with other settings may lead now */
    /* whatever happens below (std::bad_alloc included): the batch stops counting as running, and every follower that
     * has not been handed its result gets an error instead of waiting for ever */
    struct Release {
        MergeQueue& Q; std::vector<MergeReq*>& batch; MergeReq* me;
        ~Release() {
            {
                std::lock_guard<std::mutex> lock(Q.mu);
                --Q.running;
                Q.running_calls -= (int)batch.size();
                for (MergeReq* r : batch) {
                    if (r == me || r->done) continue;
                    if (!r->served) { r->rc = MI_DMRECON_EDEVICE; r->err = "the merged batch this call was part of failed"; }
                    r->done = true;
                }
            }
            Q.cv.notify_all();
        }
    } release{Q, batch, &me};
    int rc = 0;
    try {
        if (batch.size() == 1) {
            rc = reconstruct_batch(c, st, n_refs, ref_views, maps, nullptr, status_out, stats);
            if (stats) stats->n_merged_calls = 1;
            return rc;
        }
        size_t total = 0;
        for (MergeReq* r : batch) total += (size_t)r->n;
        std::vector<int32_t> refs; refs.reserve(total);
        std::vector<mi_dmrecon_maps> mm; mm.reserve(total);
        std::vector<int32_t> status(total, 0);
        for (MergeReq* r : batch) { refs.insert(refs.end(), r->refs, r->refs + r->n); mm.insert(mm.end(), r->maps, r->maps + r->n); }
        mi_dmrecon_stats bs;
        std::memset(&bs, 0, sizeof(bs));
        const int brc = reconstruct_batch(c, st, (int32_t)total, refs.data(), mm.data(), nullptr, status.data(), &bs);
        const std::string berr = brc ? g_err : std::string();
        bs.n_merged_calls = (int64_t)batch.size();
        /* a failure of the batch as a whole (device error, bad argument) is everybody's; anything else is per view */
        const bool whole = brc == MI_DMRECON_EDEVICE || brc == MI_DMRECON_EINVAL;
        size_t off = 0;
        for (MergeReq* r : batch) {
            int first = 0, n_ok = 0;
            for (int i = 0; i < r->n; ++i) {
                const int sv = whole ? brc : status[off + i];
                if (r->status) r->status[i] = sv;
                if (sv == 0) ++n_ok; else if (!first) first = sv;
            }
            if (whole) { r->rc = brc; r->err = berr; }
            else if (n_ok == 0) {
                r->rc = first;
                r->err = first == MI_DMRECON_ECANCELLED ? "cancelled" : first == MI_DMRECON_EFOOTPRINT ? "Negative pixel footprint"
                       : first == MI_DMRECON_EGVS ? "Global View Selection failed"
                       : first == MI_DMRECON_ENOIMAGE ? "a selected neighbour view has no image" : "reconstruction failed";
            } else r->rc = 0;
            if (r->stats) {
                if (r == &me) *r->stats = bs;
                else { std::memset(r->stats, 0, sizeof(*r->stats)); r->stats->merged_into_other_call = 1; }
            }
            r->served = true;
            off += (size_t)r->n;
        }
        rc = me.rc;
        if (rc) g_err = me.err;
    } catch (const std::bad_alloc&) {
        rc = fail(MI_DMRECON_EDEVICE, "out of host memory");
    }
    return rc;
}

static int mi_dmrecon_patch_optimize_impl(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view, int32_t n,
                              const int32_t* xy, const float* hyp, const int32_t* local, int32_t lanes_per_view,
                              float* out, int32_t* out_local) {
    if (!c || !xy || !hyp || !out || !out_local || n < 0) return fail(MI_DMRECON_EINVAL, "null argument");
    int rc = check_settings(st);
    if (rc) return rc;
    const MiDeviceApi& D = *mi_device_api(st->filterWidth);
    HIP_TRY(hipSetDevice(c->device));
    JobHost jh; jh.ref_view = ref_view;
    rc = plan_global_views(c, st, ref_view, jh.global);
    if (rc) return rc;
    if (jh.global.empty()) return fail(MI_DMRECON_EGVS, "Global View Selection failed");
    rc = sync_views(c);
    if (rc) return rc;
    HostLevel const& L = c->sc->views[ref_view].levels[st->scale];
    jh.w = L.w; jh.h = L.h;
    std::vector<JobHost> jobs(1, jh); JobVec dj(1);
    fill_job(c, st, jobs[0], dj[0]);
    size_t total_px = 0;
    rc = alloc_maps(c, jobs, dj, total_px, (size_t)std::max(n, 0), st->nrReconNeighbors);
    if (rc) return rc;
    if (n == 0) return 0;
    std::vector<DevEntry> ent(n); std::vector<DevHyp> hy(n); std::vector<uint32_t> hyx(2 * (size_t)n, 0xFFFFFFFFu);
    const int nch = mi_dmrecon_local_view_channels(st->nrReconNeighbors);
    for (int i = 0; i < n; ++i) {
        ent[i].job = 0;
        int x = xy[2 * i], y = xy[2 * i + 1];
        if (x < 0 || y < 0 || x >= L.w || y >= L.h) { x = 0; y = 0; }     /* fails the border test -> conf 0 */
        ent[i].xy = x | (y << 16);
        hy[i].depth = hyp[3 * i]; hy[i].dzI = hyp[3 * i + 1]; hy[i].dzJ = hyp[3 * i + 2];
        unsigned long long packed[2] = {0, 0}; int cnt = 0;
        for (int k = 0; k < nch; ++k) {
            int id = local ? local[nch * i + k] : -1;
            if (id < 0) continue;
            std::vector<int>::const_iterator it = std::lower_bound(jh.global.begin(), jh.global.end(), id);
            if (it == jh.global.end() || *it != id) return fail(MI_DMRECON_EINVAL, "local view %d is not a global view", id);
            packed[cnt >> 3] |= (unsigned long long)(it - jh.global.begin()) << (8 * (cnt & 7));
            ++cnt;
        }
        for (; cnt < 16; ++cnt) packed[cnt >> 3] |= (unsigned long long)MI_VIEW_NONE << (8 * (cnt & 7));
        hy[i].views = (uint32_t)packed[0]; hy[i].views_hi = (uint32_t)(packed[0] >> 32);
        if (nch == 16) { hyx[2 * (size_t)i] = (uint32_t)packed[1]; hyx[2 * (size_t)i + 1] = (uint32_t)(packed[1] >> 32); }
    }
    if (nch == 16) {
        /* view slots 8..15 of the propagated sets: the third part of the batch's d_xviews (alloc_maps) */
        uint32_t* d_hx = dj[0].results_x + 2 * std::max(total_px, (size_t)n);
        dj[0].hyp_x = d_hx;
        HIP_TRY(hipMemcpyAsync(d_hx, hyx.data(), 2 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    if (c->bs.d_jobs.reserve(1) || c->bs.d_work.reserve(n) || c->bs.d_hyp.reserve(n) || c->bs.d_results.reserve(n))
        return fail(MI_DMRECON_EDEVICE, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(c->bs.d_jobs.p, dj.data(), sizeof(DevJob), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->bs.d_work.p, ent.data(), n * sizeof(DevEntry), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->bs.d_hyp.p, hy.data(), n * sizeof(DevHyp), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    /* lanes_per_view = 16 runs the hook through the latency layout, anything else through the throughput layout */
    const int lpv = lanes_per_view == 16 ? 16 : 1;
    if (lpv == 16 && st->nrReconNeighbors > 8)
        return fail(MI_DMRECON_EINVAL, "more than eight local views run in the throughput layout only (lanes_per_view = 1)");
    const unsigned ppw = lpv == 16 ? 1u : patches_per_wave(st);
    D.optimize(c->stream, lpv, ((unsigned)n + ppw - 1) / ppw, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, dev_settings(st),
               c->bs.d_work.p, c->bs.d_hyp.p, c->bs.d_results.p, nullptr, (unsigned)n, 0u, 0xFFFFFFFFu, 0, c->d_counters,
               nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr);
    HIP_TRY(hipGetLastError());
    std::vector<DevResult> res(n); std::vector<uint32_t> resx(2 * (size_t)n, 0xFFFFFFFFu);
    HIP_TRY(hipMemcpyAsync(res.data(), c->bs.d_results.p, n * sizeof(DevResult), hipMemcpyDeviceToHost, c->stream));
    if (nch == 16) HIP_TRY(hipMemcpyAsync(resx.data(), dj[0].results_x, 2 * (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c->stream));
    for (int i = 0; i < n; ++i) {
        float* o = out + 8 * i;
        o[0] = res[i].conf; o[1] = res[i].depth; o[2] = res[i].dzI; o[3] = res[i].dzJ;
        o[4] = res[i].nx; o[5] = res[i].ny; o[6] = res[i].nz; o[7] = (float)res[i].iters;
        for (int k = 0; k < nch; ++k) {
            const unsigned word = k < 4 ? res[i].views : k < 8 ? res[i].views_hi : resx[2 * (size_t)i + ((k - 8) >> 2)];
            const unsigned g = (word >> (8 * (k & 3))) & 0xFFu;
            out_local[nch * i + k] = (g == MI_VIEW_NONE || g >= jh.global.size()) ? -1 : jh.global[g];
        }
    }
    return 0;
}

static int mi_dmrecon_patch_eval_impl(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view, int32_t x, int32_t y,
                          float depth, float dzI, float dzJ, float* master, float* ncc, int32_t* ok, float* col,
                          float* deriv, int32_t* level) {
    if (!c || !master || !ncc || !ok || !col || !deriv || !level) return fail(MI_DMRECON_EINVAL, "null argument");
    int rc = check_settings(st);
    if (rc) return rc;
    const MiDeviceApi& D = *mi_device_api(st->filterWidth);
    HIP_TRY(hipSetDevice(c->device));
    JobHost jh; jh.ref_view = ref_view;
    rc = plan_global_views(c, st, ref_view, jh.global);
    if (rc) return rc;
    if (jh.global.empty()) return fail(MI_DMRECON_EGVS, "Global View Selection failed");
    rc = sync_views(c);
    if (rc) return rc;
    HostLevel const& L = c->sc->views[ref_view].levels[st->scale];
    jh.w = L.w; jh.h = L.h;
    std::vector<JobHost> jobs(1, jh); JobVec dj(1);
    fill_job(c, st, jobs[0], dj[0]);
    size_t total_px = 0;
    rc = alloc_maps(c, jobs, dj, total_px, 0, st->nrReconNeighbors);
    if (rc) return rc;
    const int G = (int)jh.global.size();
    const size_t NS3 = 3 * (size_t)st->filterWidth * st->filterWidth;       /* floats per view: fw x fw samples, 3 channels */
    const size_t nfl = 5 + G + 2 * (size_t)G * NS3;
    DevBuf<float> dout; DevBuf<int32_t> diout;
    if (dout.reserve(nfl) || diout.reserve(2 * G) || c->bs.d_jobs.reserve(1)) return fail(MI_DMRECON_EDEVICE, "hipMalloc failed");
    HIP_TRY(hipMemsetAsync(dout.p, 0, nfl * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(diout.p, 0, 2 * G * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemcpyAsync(c->bs.d_jobs.p, dj.data(), sizeof(DevJob), hipMemcpyHostToDevice, c->stream));
    float* d_master = dout.p; float* d_ncc = dout.p + 5; float* d_col = d_ncc + G; float* d_der = d_col + (size_t)G * NS3;
    D.patch_eval(c->stream, c->bs.d_jobs.p, c->sc->d_views.p, c->sc->d_lut, dev_settings(st), x, y, depth, dzI, dzJ,
                         d_master, d_ncc, diout.p, d_col, d_der, diout.p + G);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(master, d_master, 5 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(ncc, d_ncc, G * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(col, d_col, (size_t)G * NS3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(deriv, d_der, (size_t)G * NS3 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(ok, diout.p, G * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(level, diout.p + G, G * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c->stream));
    dout.release(); diout.release();
    return G;
}

/* apps/scene2pset per-view body (scene2pset.cc:262-356); kernels in pointset_device.hip */
static int mi_dmrecon_pointset_impl(mi_dmrecon_ctx* c, const mi_dmrecon_camera* cam, int32_t w, int32_t h, const float* depth,
                        const uint8_t* color, int32_t color_channels, const mi_dmrecon_pointset_options* opt,
                        int32_t capacity, int32_t* pixel, float* pos, float* normal, float* color_out, float* scale,
                        float* conf, int32_t* n_out) {
    if (!c || !cam || !depth || !n_out) return fail(MI_DMRECON_EINVAL, "pointset: null argument");
    if (w < 2 || h < 2) { *n_out = 0; return w < 0 || h < 0 ? fail(MI_DMRECON_EINVAL, "pointset: bad size") : 0; }
    if (color && color_channels != 1 && color_channels != 3) return fail(MI_DMRECON_EINVAL, "pointset: 1 or 3 colour channels");
    if (capacity < 0) return fail(MI_DMRECON_EINVAL, "pointset: negative capacity");
    if (cam->flen == 0.f) return fail(MI_DMRECON_EINVAL, "pointset: invalid camera");     /* scene2pset.cc:276 */
    HIP_TRY(hipSetDevice(c->device));
    PsParams P;
    P.w = w; P.h = h;
    float K[9];
    calibration(*cam, (float)w, (float)h, K, P.inv);
    for (int i = 0; i < 3; ++i) {                              /* CameraInfo::fill_cam_to_world, camera.cc:83-93 */
        for (int j = 0; j < 3; ++j) P.ctw[4 * i + j] = cam->rot[3 * j + i];
        P.ctw[4 * i + 3] = -((cam->rot[i] * cam->trans[0] + cam->rot[3 + i] * cam->trans[1]) + cam->rot[6 + i] * cam->trans[2]);
    }
    P.dd_factor = opt ? opt->dd_factor : 5.0f;
    P.scale_factor = opt ? opt->scale_factor : 2.5f;
    P.conf_iterations = opt ? opt->conf_iterations : 4;
    if (P.conf_iterations < 1) return fail(MI_DMRECON_EINVAL, "pointset: conf_iterations < 1");
    if (P.conf_iterations > 127) return fail(MI_DMRECON_EINVAL, "pointset: conf_iterations > 127 (border distances are kept in 8 bits)");
    const size_t npix = (size_t)w * h;
    struct Tmp {                                               /* freed on every return path */
        DevBuf<float> depth; DevBuf<uint8_t> cells; DevBuf<PsVertex> verts;
        ~Tmp() { depth.release(); cells.release(); verts.release(); }
    } tmp;
    DevBuf<float>& d_depth = tmp.depth; DevBuf<uint8_t>& d_cells = tmp.cells; DevBuf<PsVertex>& d_verts = tmp.verts;
    if (d_depth.reserve(npix) || d_cells.reserve(npix) || d_verts.reserve(npix)) return fail(MI_DMRECON_EDEVICE, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(d_depth.p, depth, npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
    mi_ps_launch(c->stream, P, d_depth.p, d_cells.p, d_verts.p);
    HIP_TRY(hipGetLastError());
    std::vector<PsVertex> hv(npix);
    HIP_TRY(hipMemcpyAsync(hv.data(), d_verts.p, npix * sizeof(PsVertex), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c->stream));
    int32_t n = 0;
    const float inv_it = (float)P.conf_iterations;
    for (size_t i = 0; i < npix; ++i) {
        PsVertex const& v = hv[i];
        if (!v.used) continue;
        if (n < capacity) {
            if (pixel) pixel[n] = (int32_t)i;
            if (pos) { pos[3 * n] = v.pos[0]; pos[3 * n + 1] = v.pos[1]; pos[3 * n + 2] = v.pos[2]; }
            if (normal) { normal[3 * n] = v.nrm[0]; normal[3 * n + 1] = v.nrm[1]; normal[3 * n + 2] = v.nrm[2]; }
            if (scale) scale[n] = v.scale;
            if (conf) conf[n] = v.level < 0 ? 1.0f : (float)v.level / inv_it;
            if (color_out) {                                      /* depthmap.cc:299-307: bytes / 255 */
                for (int k = 0; k < 3; ++k)
                    color_out[3 * n + k] = color ? (float)color[i * color_channels + (color_channels == 3 ? k : 0)] / 255.0f : 0.f;
            }
        }
        ++n;
    }
    *n_out = n;
    return 0;
}

/* test hook (include/mi_dmrecon_debug.h): the reference view with this id gets a negative pixel footprint in the calls
 * that follow (-1: none), see fill_job */
void mi_dmrecon_debug_inject_footprint(int view_id) { g_inject_footprint.store(view_id); }

/* Test hook (include/mi_dmrecon_debug.h): the block map of a front launch with teams (build_front_teams) for n_views views on
 * a device of n_cus compute units; map_out gets min(grid, cap) words.  Needs no GPU. */
int mi_dmrecon_debug_front_teams(int32_t n_views, int32_t n_cus, int32_t want, const int64_t* empty, uint32_t* map_out, int32_t cap,
                                 int32_t* grid_out, int32_t* team_min_out, int32_t* team_max_out) {
    try {
        std::vector<long long> e;
        if (empty) e.assign(empty, empty + std::max(n_views, 0));
        const FrontTeams ft = build_front_teams(n_views, n_cus, want, empty ? e.data() : nullptr);
        if (grid_out) *grid_out = (int32_t)ft.grid;
        if (team_min_out) *team_min_out = ft.team_min;
        if (team_max_out) *team_max_out = ft.team_max;
        for (size_t i = 0; map_out && i < ft.map.size() && (int32_t)i < cap; ++i) map_out[i] = ft.map[i];
        return 0;
    } catch (std::exception const& ex) { return fail(MI_DMRECON_EDEVICE, "%s", ex.what()); }
}

/* Test hook (include/mi_dmrecon_debug.h): the HOST half of the planning -- global view selection of reference view `ref`,
 * exactly the code a reconstruct call runs (plan_global_views: from the scene tables, or directly with tables = 0) -- on
 * cameras and features alone, so that it can be checked against the oracle without a GPU.  ms_out (optional): the time
 * of `repeats` selections, the scene tables already built.  n_seeds_out (optional): also the view's seeds. */
int mi_dmrecon_debug_plan_views_host(int32_t n_views, const mi_dmrecon_camera* cams, const int32_t* widths, const int32_t* heights,
                                     int32_t n_feat, const float* pos, const int32_t* off, const int32_t* ids,
                                     const mi_dmrecon_settings* st, int32_t ref, int32_t tables, int32_t repeats,
                                     int32_t* ids_out, int32_t* n_out, double* ms_out,
                                     int32_t seed_cap, int32_t* seed_xy_out, float* seed_depth_out, int32_t* n_seeds_out) {
    try {
        if (n_views <= 0 || !cams || !widths || !heights || !st || !ids_out || !n_out) return fail(MI_DMRECON_EINVAL, "null argument");
        mi_dmrecon_ctx c;
        c.device = -1;
        c.sc = std::make_shared<SceneStore>();
        c.sc->views.resize(n_views);
        for (int i = 0; i < n_views; ++i) (void)host_view_set_camera(c.sc->views[i], &cams[i], widths[i], heights[i]);
        c.sc->features.resize(n_feat);
        for (int i = 0; i < n_feat; ++i) {
            Feature& f = c.sc->features[i];
            f.pos[0] = pos[3 * i]; f.pos[1] = pos[3 * i + 1]; f.pos[2] = pos[3 * i + 2];
            f.ref_begin = off[i]; f.ref_end = off[i + 1];
        }
        c.sc->feat_refs.assign(ids, ids + (n_feat ? off[n_feat] : 0));
        build_features_by_view(*c.sc);
        if (!tables) { c.sc->geom.built = true; c.sc->geom.has_plx = false; }   /* as for a bundle too large for the tables */
        std::vector<int> global;
        int rc = plan_global_views(&c, st, ref, global);                        /* (builds the tables) */
        const double t0 = now_ms();
        for (int k = 1; k < repeats && rc == 0; ++k) rc = plan_global_views(&c, st, ref, global);
        if (ms_out) *ms_out = now_ms() - t0;
        if (rc) return rc;
        *n_out = (int32_t)global.size();
        for (size_t i = 0; i < global.size(); ++i) ids_out[i] = global[i];
        if (n_seeds_out) {
            /* ... and the seeds of the view (plan_seeds: the host half of processFeatures), in feature order */
            JobHost job;
            job.ref_view = ref; job.global = global;
            plan_seeds(&c, st, job, 0);
            *n_seeds_out = (int32_t)job.seeds.size();
            for (size_t i = 0; i < job.seeds.size() && (int32_t)i < seed_cap; ++i) {
                if (seed_xy_out) { seed_xy_out[2 * i] = (int32_t)(job.seeds[i].xy & 0xFFFF); seed_xy_out[2 * i + 1] = (int32_t)(job.seeds[i].xy >> 16); }
                if (seed_depth_out) seed_depth_out[i] = job.seed_hyp[i].depth;
            }
        }
        return 0;
    } catch (std::exception const& e) { return fail(MI_DMRECON_EDEVICE, "%s", e.what()); }
}

/* test hook (include/mi_dmrecon_debug.h): the scratch sets of the context's scene that no call holds at the moment, and the
 * pixel capacity of the largest of them (ScratchLease) */
int mi_dmrecon_debug_scratch_sets(mi_dmrecon_ctx* c, long long* pixels_max) {
    if (!c) return -1;
    std::lock_guard<std::mutex> lock(c->sc->pool_mu);
    size_t best = 0;
    for (const BatchScratch& b : c->sc->scratch_pool) best = std::max(best, b.pixels());
    if (pixels_max) *pixels_max = (long long)best;
    return (int)c->sc->scratch_pool.size();
}

/* development aid (include/mi_dmrecon_debug.h): the debug buffer of MI_PROBE builds (tools/patch_probe.py).  The first
 * call allocates `n` words on the device; later calls copy up to n words out and clear the buffer. */
int mi_dmrecon_debug_region_mark(mi_dmrecon_ctx* c, int tag) {
    if (!c || c->device < 0 || !c->stream) return fail(MI_DMRECON_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    mi_launch_region_mark(c->stream, (unsigned)tag);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int mi_dmrecon_debug_buffer(unsigned long long* out, int n) {
    static int cap = 0;
    if (!mi_debug_tbuf) {
        if (n <= 0 || hipMalloc((void**)&mi_debug_tbuf, (size_t)n * sizeof(unsigned long long)) != hipSuccess) return -1;
        cap = n;
        (void)hipMemset(mi_debug_tbuf, 0, (size_t)cap * 8);
        return 0;
    }
    if (out) (void)hipMemcpy(out, mi_debug_tbuf, (size_t)std::min(n, cap) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipMemset(mi_debug_tbuf, 0, (size_t)cap * 8);
    return 0;
}

/* ---- the entry points that allocate host memory: no exception crosses the C ABI (std::bad_alloc -> MI_DMRECON_EDEVICE) */
int mi_dmrecon_global_view_selection(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view, int32_t* ids_out, int32_t* n_out) {
    try { return mi_dmrecon_global_view_selection_impl(c, st, ref_view, ids_out, n_out); }
    catch (const std::bad_alloc&) { return fail(MI_DMRECON_EDEVICE, "out of host memory"); }
}

int mi_dmrecon_reconstruct(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t n_refs, const int32_t* ref_views, mi_dmrecon_maps* maps, mi_dmrecon_progress* progress, int32_t* status_out, mi_dmrecon_stats* stats) {
    /* the statistics are gathered in an object of THIS build's type and handed over in the size the caller has room for
     * (mi_dmrecon_stats::struct_size): a caller built against a shorter header is not overrun */
    mi_dmrecon_stats full;
    std::memset(&full, 0, sizeof(full));
    size_t room = 0;
    if (stats) {
        if (stats->struct_size < (int64_t)sizeof(int64_t) || stats->struct_size > (int64_t)(1 << 20))
            return fail(MI_DMRECON_EINVAL, "mi_dmrecon_stats::struct_size must be set to sizeof(mi_dmrecon_stats) before the call");
        room = std::min((size_t)stats->struct_size, sizeof(full));
    }
    int rc;
    try { rc = mi_dmrecon_reconstruct_impl(c, st, n_refs, ref_views, maps, progress, status_out, stats ? &full : nullptr); }
    catch (const std::bad_alloc&) { rc = fail(MI_DMRECON_EDEVICE, "out of host memory"); }
    if (stats) { full.struct_size = (int64_t)room; std::memcpy(stats, &full, room); }
    return rc;
}

int mi_dmrecon_patch_optimize(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view, int32_t n, const int32_t* xy, const float* hyp, const int32_t* local, int32_t lanes_per_view, float* out, int32_t* out_local) {
    try { return mi_dmrecon_patch_optimize_impl(c, st, ref_view, n, xy, hyp, local, lanes_per_view, out, out_local); }
    catch (const std::bad_alloc&) { return fail(MI_DMRECON_EDEVICE, "out of host memory"); }
}

int mi_dmrecon_patch_eval(mi_dmrecon_ctx* c, const mi_dmrecon_settings* st, int32_t ref_view, int32_t x, int32_t y, float depth, float dzI, float dzJ, float* master, float* ncc, int32_t* ok, float* col, float* deriv, int32_t* level) {
    try { return mi_dmrecon_patch_eval_impl(c, st, ref_view, x, y, depth, dzI, dzJ, master, ncc, ok, col, deriv, level); }
    catch (const std::bad_alloc&) { return fail(MI_DMRECON_EDEVICE, "out of host memory"); }
}

int mi_dmrecon_set_features(mi_dmrecon_ctx* c, int32_t n, const float* pos, const int32_t* off, const int32_t* ids) {
    try { return mi_dmrecon_set_features_impl(c, n, pos, off, ids); }
    catch (const std::bad_alloc&) { return fail(MI_DMRECON_EDEVICE, "out of host memory"); }
}

int mi_dmrecon_pointset(mi_dmrecon_ctx* c, const mi_dmrecon_camera* cam, int32_t w, int32_t h, const float* depth, const uint8_t* color, int32_t color_channels, const mi_dmrecon_pointset_options* opt, int32_t capacity, int32_t* pixel, float* pos, float* normal, float* color_out, float* scale, float* conf, int32_t* n_out) {
    try { return mi_dmrecon_pointset_impl(c, cam, w, h, depth, color, color_channels, opt, capacity, pixel, pos, normal, color_out, scale, conf, n_out); }
    catch (const std::bad_alloc&) { return fail(MI_DMRECON_EDEVICE, "out of host memory"); }
}

}  /* extern "C" */
