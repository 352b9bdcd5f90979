/*
 * mvs::DMRecon on top of the C ABI of include/mi_dmrecon.h -- the host shim that makes
 * libmi_dmrecon.so a drop-in for libmve_dmrecon.a.  Compiled against the reference's libs/mve,
 * libs/math and libs/util headers (scene I/O stays MVE's); nothing from libs/dmrecon is used.
 *
 * Behaviour kept from the reference (libs/dmrecon/dmrecon.cc):
 *   ctor  :37-46,74-75  std::invalid_argument for a bad master view / scale / embedding,
 *         :50-59        std::runtime_error if the bundle cannot be read,
 *         :81-86        "scaled image size" message unless quiet;
 *   start :101-105      a set `cancelled` flag ends in RECON_CANCELLED and writes nothing,
 *         :119-145      view->set_image for depth-L<s>, dz-L<s>, conf-L<s>, undist-L<s>,
 *         :148-162      "Filled N pixels" / "MVS took" messages,
 *         :223          std::runtime_error("Global View Selection failed") escapes to the caller.
 * What replaces ImagePyramidCache (image_pyramid.cc:98-154): a process-wide registry that uploads a
 * scene's views to each GPU once (pyramids are built on the device) and hands every host thread its
 * own forked context, so the OpenMP loop of apps/dmrecon/dmrecon.cc:285-318 keeps all GPUs busy.
 * Device choice: MI_DMRECON_DEVICES="0,2,5" (default: every visible GPU), threads round-robin.
 */
#include "dmrecon/dmrecon.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <vector>

#include "mve/image.h"
#include "mve/image_tools.h"
#include "mve/mesh_io_ply.h"
#include "mve/view.h"
#include "util/file_system.h"
#include "util/string_utils.h"
#include "mi_dmrecon.h"

MVS_NAMESPACE_BEGIN

namespace {

/* MI_DMRECON_TRACE: where the host side of a run spends its time (stderr, milliseconds since the first line) */
bool tracing() { static bool const on = std::getenv("MI_DMRECON_TRACE") != nullptr; return on; }
double trace_ms()
{
    static std::chrono::steady_clock::time_point const t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
void trace(char const* what, double since_ms, long n = -1)
{
    if (!tracing()) return;
    double const now = trace_ms();
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (n >= 0) std::fprintf(stderr, "[mvs::DMRecon shim] t=%9.2f ms  %-34s %9.2f ms  (%ld)\n", now, what, now - since_ms, n);
    else std::fprintf(stderr, "[mvs::DMRecon shim] t=%9.2f ms  %-34s %9.2f ms\n", now, what, now - since_ms);
}

/* The HIP runtime takes 90-190 ms to start (first hipGetDeviceCount).  In a program that links this shim it starts in a
 * helper thread before main() -- next to the driver's own start-up (arguments, scene directory, view headers) instead of
 * inside the first DMRecon constructor.  Joined at exit, so a run that ends at once (--help) waits for it. */
struct HipWarmUp {
    std::thread t;
    HipWarmUp() {
        char const* e = std::getenv("MI_DMRECON_WARMUP");               /* =0: a program that links the shim but may never use it */
        if (!e || std::atoi(e) != 0) t = std::thread([]() { (void)mi_dmrecon_device_count(); });
    }
    ~HipWarmUp() { if (t.joinable()) t.join(); }
} g_hip_warm_up;

[[noreturn]] void raise_from(int rc)
{
    std::string msg = mi_dmrecon_last_error();
    switch (rc) {
        case MI_DMRECON_EINVAL: throw std::invalid_argument(msg);
        case MI_DMRECON_EFOOTPRINT: throw std::out_of_range(msg);
        default: throw std::runtime_error(msg);
    }
}

/* One request of a host thread: reconstruct one reference view into caller-owned maps. */
struct Request {
    mi_dmrecon_settings st;
    int32_t ref = 0;
    mi_dmrecon_maps maps;
    mi_dmrecon_progress* prog = nullptr;       /* the requesting thread's POD (relayed to its mvs::Progress) */
    int rc = 0;
    std::string err;
    bool done = false;
};

/*
 * One GPU: the resident scene (parent context), a few forked contexts ("executors", one HIP stream each) and a queue
 * of requests.  MVE's driver runs one mvs::DMRecon per OpenMP thread (apps/dmrecon/dmrecon.cc:285-318); the library
 * is fastest when it gets many reference views per call (one set of launches, one ~600-round latency tail per batch
 * instead of per view).  So the threads' requests are batched: a thread that finds a free executor takes ALL requests
 * pending at that moment (same settings) into one mi_dmrecon_reconstruct call; the others wait for their result.
 * Under load (more OpenMP threads than executors) batches form by themselves; a lone request runs immediately.
 */
class Slot
{
public:
    int device = 0;
    mi_dmrecon_ctx* parent = nullptr;
    std::mutex parent_mu;                      /* the parent context serves short queries (level size, level image) */

    ~Slot() { release(); }

    void release()
    {
        for (std::size_t i = 0; i < executors.size(); ++i) mi_dmrecon_ctx_destroy(executors[i]);
        executors.clear(); idle.clear();
        if (parent) mi_dmrecon_ctx_destroy(parent);
        parent = nullptr;
    }

    /* instances attached to this GPU that have not called start() yet: the driver constructs and starts one DMRecon per
     * OpenMP thread at the same moment -- a thread about to run a batch gives those a moment to file their requests */
    std::atomic<int> announced{0};
    void announce() { announced.fetch_add(1); }
    void withdraw()                            /* the instance starts (its request follows at once) or goes away unstarted */
    {
        std::lock_guard<std::mutex> lk(mu);
        announced.fetch_sub(1);
        cv.notify_all();
    }

    void file(Request& r)
    {
        std::lock_guard<std::mutex> lk(mu);
        pending.push_back(&r);
        cv.notify_all();
    }

    /* returns when the request (filed before) has been served */
    void submit(Request& r)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (r.done) return;
            /* only a thread whose own request is still queued turns executor (one whose request is in flight in
             * another batch just waits for it) */
            if (std::find(pending.begin(), pending.end(), &r) != pending.end() && !gathering
                && (!idle.empty() || executors.size() < max_executors())) {
                mi_dmrecon_ctx* ex = nullptr;
                if (!idle.empty()) { ex = idle.back(); idle.pop_back(); }
                else {
                    int rc = mi_dmrecon_ctx_fork(parent, &ex);
                    if (rc != 0) { fail_all(rc); continue; }
                    executors.push_back(ex);
                }
                /* the threads that have constructed their instance are on their way (microseconds apart): without the
                 * wait the first to arrive runs a batch of ONE view and the rest a second batch behind it */
                if (announced.load() > 0 && gather_pays) {
                    /* (one gatherer at a time: the others wait for their results instead of forking executors of their
                     * own under the lock the late-comers need to file their requests) */
                    gathering = true;
                    auto const until = std::chrono::steady_clock::now() + std::chrono::microseconds(gather_us());
                    while (announced.load() > 0 && cv.wait_until(lk, until) != std::cv_status::timeout) { }
                    gathering = false;
                    /* a caller that constructs its instances up front and starts them one after the other would pay the
                     * wait for every view: after a wait that nobody joined, batches start at once until company shows up */
                    if (pending.size() <= 1) gather_pays = false;
                }
                if (pending.size() > 1) gather_pays = true;
                /* everything pending with the settings of the oldest request, up to max_batch() views */
                std::vector<Request*> batch;
                mi_dmrecon_settings const key = pending.front()->st;
                for (std::deque<Request*>::iterator it = pending.begin(); it != pending.end() && batch.size() < max_batch();) {
                    if (same_settings((*it)->st, key)) { batch.push_back(*it); it = pending.erase(it); } else ++it;
                }
                lk.unlock();
                execute(ex, batch);
                lk.lock();
                idle.push_back(ex);
                for (std::size_t i = 0; i < batch.size(); ++i) batch[i]->done = true;
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
    }

private:
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Request*> pending;
    std::vector<mi_dmrecon_ctx*> executors, idle;
    bool gathering = false;                    /* a thread is waiting for the announced instances before it runs a batch */
    bool gather_pays = true;                   /* the last wait was joined by somebody (or there has been none yet) */

    static std::size_t env_or(char const* name, std::size_t dflt)
    {
        char const* e = std::getenv(name);
        int v = e ? std::atoi(e) : 0;
        return v > 0 ? (std::size_t)v : dflt;
    }
    static std::size_t max_executors() { static std::size_t v = env_or("MI_DMRECON_EXECUTORS", 4); return v; }
    static std::size_t max_batch() { static std::size_t v = env_or("MI_DMRECON_MAX_BATCH", 128); return v; }
    static std::size_t gather_us() { static std::size_t v = env_or("MI_DMRECON_GATHER_US", 2000); return v; }

    static bool same_settings(mi_dmrecon_settings const& a, mi_dmrecon_settings const& b)
    {
        return a.filterWidth == b.filterWidth && a.minNCC == b.minNCC && a.minParallax == b.minParallax
            && a.acceptNCC == b.acceptNCC && a.minRefineDiff == b.minRefineDiff && a.maxIterations == b.maxIterations
            && a.nrReconNeighbors == b.nrReconNeighbors && a.globalVSMax == b.globalVSMax && a.scale == b.scale
            && a.useColorScale == b.useColorScale
            && std::equal(a.aabbMin, a.aabbMin + 3, b.aabbMin) && std::equal(a.aabbMax, a.aabbMax + 3, b.aabbMax);
    }

    void fail_all(int rc)                      /* mu held */
    {
        std::string msg = mi_dmrecon_last_error();
        for (std::deque<Request*>::iterator it = pending.begin(); it != pending.end(); ++it) {
            (*it)->rc = rc; (*it)->err = msg; (*it)->done = true;
        }
        pending.clear();
        cv.notify_all();
    }

    static void execute(mi_dmrecon_ctx* ex, std::vector<Request*> const& batch)
    {
        std::size_t const n = batch.size();
        std::vector<int32_t> refs(n), status(n, 0);
        std::vector<mi_dmrecon_maps> maps(n);
        std::vector<mi_dmrecon_progress> prog(n);
        for (std::size_t i = 0; i < n; ++i) { refs[i] = batch[i]->ref; maps[i] = batch[i]->maps; prog[i] = *batch[i]->prog; }
        /* progress / cancellation relay between the batch's array and the requesting threads' PODs.  The library
         * ends views individually (status[i]): a view cancelled while batched with others does not touch those. */
        std::atomic<bool> running(true);
        std::thread relay([&]() {
            while (running.load()) {
                for (std::size_t i = 0; i < n; ++i) {
                    if (batch[i]->prog->cancelled) prog[i].cancelled = 1;
                    batch[i]->prog->status = prog[i].status;
                    batch[i]->prog->filled = prog[i].filled;
                    batch[i]->prog->queueSize = prog[i].queueSize;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        });
        mi_dmrecon_stats stats;
        stats.struct_size = (int64_t)sizeof(stats);           /* the library fills what this build has room for */
        double const t_call = trace_ms();
        int rc = mi_dmrecon_reconstruct(ex, &batch[0]->st, (int32_t)n, refs.data(), maps.data(), prog.data(), status.data(), &stats);
        std::string msg = rc != 0 ? mi_dmrecon_last_error() : "";
        trace("batch reconstructed (views)", t_call, (long)n);
        running.store(false);
        relay.join();
        for (std::size_t i = 0; i < n; ++i) {
            Request& q = *batch[i];
            q.prog->status = prog[i].status;
            if (status[i] != 0) {              /* this view's own outcome; the others ran (mi_dmrecon.h) */
                q.rc = status[i];
                q.err = status[i] == MI_DMRECON_EGVS ? "Global View Selection failed"
                      : status[i] == MI_DMRECON_EFOOTPRINT ? "Negative pixel footprint"
                      : status[i] == MI_DMRECON_ECANCELLED ? "cancelled"
                      : status[i] == MI_DMRECON_ENOIMAGE ? "a selected neighbour view has no image" : "reconstruction failed";
            } else if (rc != 0) { q.rc = rc; q.err = msg; }
        }
    }
};

/*
 * The scene as the GPUs hold it: one Slot per GPU named in MI_DMRECON_DEVICES (default: all), one resident copy of
 * the scene on each.  A generation belongs to one (scene, embedding) pair and OWNS a reference to the mve::Scene,
 * so a freed scene's address cannot come back as a "cached" key; every mvs::DMRecon holds its generation by
 * shared_ptr, so a change of scene never destroys contexts that still have requests in flight -- the old
 * generation goes when its last DMRecon does (the reference: ImagePyramidCache hands out shared_ptrs,
 * image_pyramid.cc:98-132).
 */
struct Generation {
    mve::Scene::Ptr scene;
    std::string embedding;
    std::vector<std::unique_ptr<Slot> > slots;
    std::once_flag uploaded;
    std::exception_ptr upload_error;
    /* a view whose image could not be decoded (util::Exception of the codec): kept per view and raised where the reference
     * would have met it -- in the constructor of a DMRecon whose master view it is (loadColorImage, dmrecon.cc:79), in
     * start() of one that could select it as a neighbour (dmrecon.cc:240) -- not in everybody's constructor */
    std::vector<std::exception_ptr> decode_error;
    std::atomic<std::size_t> next_slot{0};
};

class Registry
{
public:
    static Registry& get() { static Registry r; return r; }

    /* the generation of (scene, embedding): its GPU slots exist, the scene may still be on its way (make_resident) */
    std::shared_ptr<Generation> generation_for(mve::Scene::Ptr scene, std::string const& embedding)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!current || current->scene != scene || current->embedding != embedding) {
            /* like ImagePyramidCache: one scene/embedding at a time; a different one re-uploads */
            current = std::make_shared<Generation>();
            current->scene = scene;
            current->embedding = embedding;
            init_devices(*current);
        }
        return current;
    }

    /* every GPU's copy resident: the first caller stages the scene, the others of this generation wait for it
     * (outside the registry lock) */
    static void make_resident(Generation& g)
    {
        std::call_once(g.uploaded, [&]() {
            try { upload(g); } catch (...) { g.upload_error = std::current_exception(); }
        });
        if (g.upload_error) std::rethrow_exception(g.upload_error);
    }

private:
    std::mutex mu;
    std::shared_ptr<Generation> current;

    static void init_devices(Generation& g)
    {
        int n = mi_dmrecon_device_count();
        if (n <= 0) throw std::runtime_error("mvs::DMRecon (MI355X build): no HIP device available; there is no CPU path");
        std::vector<int> devices;
        char const* env = std::getenv("MI_DMRECON_DEVICES");
        if (env && *env) {
            std::stringstream ss(env);
            std::string tok;
            while (std::getline(ss, tok, ',')) {
                int d = std::atoi(tok.c_str());
                if (d >= 0 && d < n) devices.push_back(d);
            }
        }
        if (devices.empty()) for (int d = 0; d < n; ++d) devices.push_back(d);
        for (std::size_t i = 0; i < devices.size(); ++i) {
            g.slots.push_back(std::unique_ptr<Slot>(new Slot()));
            g.slots.back()->device = devices[i];
        }
    }

    /*
     * SingleView::create for every usable view (dmrecon.cc:62-71) + ensureImages (image_pyramid.cc:55-95), for all
     * GPUs at once.  The reference decodes and down-samples under one global mutex (image_pyramid.cc:102); here the
     * views are decoded ONCE, by several host threads in parallel, and each decoded image is handed to every GPU's
     * asynchronous staging path (mi_dmrecon_set_view_async: copy + RGBA pack + pyramid kernels are only enqueued),
     * so the uploads of the GPUs and the decoding of the next views overlap.
     */
    static void upload(Generation& g)
    {
        std::size_t const ns = g.slots.size();
        double t_mark = trace_ms();
        std::vector<mi_dmrecon_ctx*> ctxs(ns, nullptr);
        std::vector<std::unique_ptr<std::mutex> > ctx_mu;
        for (std::size_t s = 0; s < ns; ++s) ctx_mu.push_back(std::unique_ptr<std::mutex>(new std::mutex()));
        /* Three things that do not need each other run side by side: HIP start-up + one context per GPU (~30 ms), MVE's
         * parse of the bundle file (55 ms alone; the reference's constructor does it first, every OpenMP thread of the
         * driver at once), and the PNG decoders, which only need a context when their first image is ready. */
        int ctx_rc = 0;
        std::string ctx_msg;
        std::mutex ready_mu;
        std::condition_variable ready_cv;
        bool ctx_ready = false;
        std::thread ctx_thread([&]() {
            double const t0 = trace_ms();
            int rc = 0;
            for (std::size_t s = 0; s < ns && rc == 0; ++s) {
                rc = mi_dmrecon_ctx_create(g.slots[s]->device, &ctxs[s]);
                if (rc != 0) ctx_msg = mi_dmrecon_last_error();
            }
            trace("contexts created (HIP start-up)", t0, (long)ns);
            std::lock_guard<std::mutex> lock(ready_mu);
            ctx_rc = rc; ctx_ready = true;
            ready_cv.notify_all();
        });
        std::exception_ptr bundle_exc;
        std::thread bundle_thread([&]() {
            double const t0 = trace_ms();
            try { g.scene->get_bundle(); }
            catch (std::exception& e) {                       /* dmrecon.cc:50-59 */
                bundle_exc = std::make_exception_ptr(std::runtime_error(std::string("Error reading bundle file: ") + e.what()));
            } catch (...) { bundle_exc = std::current_exception(); }
            trace("bundle parsed", t0);
        });
        auto wait_for_contexts = [&]() -> bool {
            std::unique_lock<std::mutex> lock(ready_mu);
            ready_cv.wait(lock, [&] { return ctx_ready; });
            return ctx_rc == 0;
        };
        mve::Scene::ViewList const& views(g.scene->get_views());
        /* A decoded image stays alive until the copies enqueued from it have run (mi_dmrecon_sync).  The views go
         * through in windows of a few per decoding thread: decode + enqueue in parallel, then one sync of every GPU,
         * then the window's images are released -- the host holds one window of decoded images, not the scene.
         * An exception of a view's decoder (util::Exception for a missing / corrupt image) must not leave its thread:
         * it is kept with the view (Generation::decode_error; the view is then simply not resident) and raised to the
         * DMRecon instances that would have met it in the reference. */
        int failed_rc = 0;
        std::string failed_msg;
        g.decode_error.assign(views.size(), std::exception_ptr());
        std::mutex err_mu;
        /* The decoders are plain threads, not an OpenMP team: this runs inside the driver's own parallel region
         * (apps/dmrecon/dmrecon.cc:285), where a nested `omp parallel` gets ONE thread -- the 20 PNGs of a scene were
         * decoded one after the other (0.64 s of the drop-in binary's 1.7 s on C3). */
        int const n_threads = (int)std::max<std::size_t>(1, std::min<std::size_t>(env_threads(), views.size()));
        std::size_t const window = (std::size_t)n_threads * 2;
        std::vector<mve::ByteImage::Ptr> keep(views.size());
        for (std::size_t base = 0; base < views.size() && failed_rc == 0; base += window) {
            std::size_t const end = std::min(views.size(), base + window);
            std::atomic<std::size_t> next(base);
            auto work = [&]() {
                for (;;) {
                    std::size_t const i = next.fetch_add(1);
                    if (i >= end) return;
                    try {
                        if (views[i] == nullptr || !views[i]->is_camera_valid()
                            || !views[i]->has_image(g.embedding, mve::IMAGE_TYPE_UINT8))
                            continue;
                        mve::ByteImage::Ptr img = views[i]->get_byte_image(g.embedding);  /* decode: per view, no shared state */
                        if (img == nullptr) continue;
                        keep[i] = img;
                        if (!wait_for_contexts()) return;
                        mve::CameraInfo const& cam = views[i]->get_camera();
                        mi_dmrecon_camera mc;
                        mc.flen = cam.flen; mc.paspect = cam.paspect;
                        mc.ppoint[0] = cam.ppoint[0]; mc.ppoint[1] = cam.ppoint[1];
                        for (int k = 0; k < 9; ++k) mc.rot[k] = cam.rot[k];
                        for (int k = 0; k < 3; ++k) mc.trans[k] = cam.trans[k];
                        for (std::size_t s = 0; s < ns; ++s) {
                            std::lock_guard<std::mutex> lock(*ctx_mu[s]);                 /* a context is not thread-safe */
                            int rc = mi_dmrecon_set_view_async(ctxs[s], (int32_t)i, &mc, img->width(), img->height(), img->channels(),
                                                               img->get_data_pointer());
                            if (rc != 0) {
                                std::lock_guard<std::mutex> elock(err_mu);
                                if (failed_rc == 0) { failed_rc = rc; failed_msg = mi_dmrecon_last_error(); }
                            }
                        }
                        views[i]->cache_cleanup();
                    } catch (...) {
                        g.decode_error[i] = std::current_exception();     /* (one writer per view) */
                        /* The reference still has a SingleView for this view (valid camera, an image of the embedding whose
                         * header it could read: dmrecon.cc:62-79) -- a candidate of everybody's global view selection -- and
                         * only meets the broken image when a reconstruction SELECTS it (dmrecon.cc:236-240).  Registered with
                         * its camera and size only: the library's selections see it, and a view that selects it gets
                         * MI_DMRECON_ENOIMAGE, which start() turns back into this exception. */
                        try {
                            mve::View::ImageProxy const* px = views[i]->get_image_proxy(g.embedding);
                            if (px != nullptr && px->width > 1 && px->height > 1 && wait_for_contexts()) {
                                mve::CameraInfo const& cam = views[i]->get_camera();
                                mi_dmrecon_camera mc;
                                mc.flen = cam.flen; mc.paspect = cam.paspect;
                                mc.ppoint[0] = cam.ppoint[0]; mc.ppoint[1] = cam.ppoint[1];
                                for (int k = 0; k < 9; ++k) mc.rot[k] = cam.rot[k];
                                for (int k = 0; k < 3; ++k) mc.trans[k] = cam.trans[k];
                                for (std::size_t s = 0; s < ns; ++s) {
                                    std::lock_guard<std::mutex> lock(*ctx_mu[s]);
                                    (void)mi_dmrecon_set_view(ctxs[s], (int32_t)i, &mc, px->width, px->height, 3, nullptr);
                                }
                            }
                        } catch (...) { }
                    }
                }
            };
            std::vector<std::thread> pool;
            int const n_here = (int)std::min<std::size_t>((std::size_t)n_threads, end - base);
            for (int t = 1; t < n_here; ++t) pool.emplace_back(work);
            work();
            for (std::size_t t = 0; t < pool.size(); ++t) pool[t].join();
            if (!wait_for_contexts()) break;
            for (std::size_t s = 0; s < ns; ++s) {
                int rc = mi_dmrecon_sync(ctxs[s]);
                if (rc != 0 && failed_rc == 0) { failed_rc = rc; failed_msg = mi_dmrecon_last_error(); }
            }
            for (std::size_t i = base; i < end; ++i) keep[i].reset();
        }
        ctx_thread.join();
        bundle_thread.join();
        auto destroy_contexts = [&]() { for (std::size_t s = 0; s < ns; ++s) if (ctxs[s]) mi_dmrecon_ctx_destroy(ctxs[s]); };
        if (bundle_exc) { destroy_contexts(); std::rethrow_exception(bundle_exc); }     /* the reference's first failure */
        if (ctx_rc != 0) {
            destroy_contexts();
            switch (ctx_rc) {
                case MI_DMRECON_EINVAL: throw std::invalid_argument(ctx_msg);
                default: throw std::runtime_error(ctx_msg);
            }
        }
        trace("views decoded + staged", t_mark, (long)views.size()); t_mark = trace_ms();
        mve::Bundle::Features const& feats = g.scene->get_bundle()->get_features();
        std::vector<float> pos(feats.size() * 3);
        std::vector<int32_t> off(feats.size() + 1, 0), ids;
        for (std::size_t i = 0; i < feats.size(); ++i) {
            for (int k = 0; k < 3; ++k) pos[3 * i + k] = feats[i].pos[k];
            for (std::size_t j = 0; j < feats[i].refs.size(); ++j) ids.push_back(feats[i].refs[j].view_id);
            off[i + 1] = (int32_t)ids.size();
        }
        if (ids.empty()) ids.push_back(0);
        for (std::size_t s = 0; s < ns && failed_rc == 0; ++s) {
            int rc = mi_dmrecon_set_features(ctxs[s], (int32_t)feats.size(), pos.data(), off.data(), ids.data());
            if (rc != 0) { failed_rc = rc; failed_msg = mi_dmrecon_last_error(); }
        }
        if (failed_rc != 0) {
            for (std::size_t s = 0; s < ns; ++s) mi_dmrecon_ctx_destroy(ctxs[s]);
            switch (failed_rc) {
                case MI_DMRECON_EINVAL: throw std::invalid_argument(failed_msg);
                default: throw std::runtime_error(failed_msg);
            }
        }
        for (std::size_t s = 0; s < ns; ++s) g.slots[s]->parent = ctxs[s];
        trace("features set", t_mark, (long)feats.size());
    }

    static std::size_t env_threads()
    {
        char const* e = std::getenv("MI_DMRECON_DECODE_THREADS");
        int v = e ? std::atoi(e) : 0;
        if (v > 0) return (std::size_t)v;
        unsigned const hw = std::thread::hardware_concurrency();
        return (std::size_t)std::max(1u, std::min(32u, hw ? hw : 16u));
    }
};

/* what a DMRecon instance keeps: its generation (alive as long as the instance) and its GPU of that generation */
struct Attachment {
    std::shared_ptr<Generation> gen;
    Slot* slot = nullptr;
    bool announced = false;                    /* counted in slot->announced until the first start() */
    void withdraw() { if (announced) { announced = false; slot->withdraw(); } }
    ~Attachment() { withdraw(); }
};

}  // namespace

DMRecon::DMRecon(mve::Scene::Ptr _scene, Settings const& _settings)
    : scene(_scene), settings(_settings), slot(), width(0), height(0)
{
    mve::Scene::ViewList const& mve_views(scene->get_views());
    if (settings.refViewNr >= mve_views.size())
        throw std::invalid_argument("Master view index out of bounds");
    if (settings.scale < 0.f)
        throw std::invalid_argument("Invalid scale factor");
    if (settings.imageEmbedding.empty())
        throw std::invalid_argument("Invalid image embedding");
    double const t_ctor = trace_ms();

    /* requests are dealt round-robin over the GPUs of the generation (schedule(dynamic,1) over views, apps/dmrecon/
     * dmrecon.cc:285, hands consecutive views to whichever thread is free: any static thread -> GPU map would do) */
    std::shared_ptr<Attachment> att = std::make_shared<Attachment>();
    att->gen = Registry::get().generation_for(scene, settings.imageEmbedding);
    att->slot = att->gen->slots[att->gen->next_slot++ % att->gen->slots.size()].get();
    /* announced BEFORE the wait for the scene: the driver's threads all sit in that wait while the first of them stages
     * the views, so by the time it ends every one of them is counted and their start() calls form one batch */
    att->slot->announce(); att->announced = true;
    this->slot = att;
    /* the master view's own validity costs nothing (dmrecon.cc:62-75: SingleView::create decodes no image): checked before
     * anything is staged -- after the bundle, whose error comes first in the reference (dmrecon.cc:50-59) */
    mve::View::Ptr ref = mve_views[settings.refViewNr];
    if (ref == nullptr || !ref->is_camera_valid()
        || !ref->has_image(settings.imageEmbedding, mve::IMAGE_TYPE_UINT8)) {
        try { this->scene->get_bundle(); }
        catch (std::exception& e) { throw std::runtime_error(std::string("Error reading bundle file: ") + e.what()); }
        throw std::invalid_argument("Invalid master view");
    }
    /* stages the scene on first use; the bundle file is read in there, next to the image decoders (dmrecon.cc:50-59:
     * std::runtime_error "Error reading bundle file: ..." -- before the master view is looked at, as in the reference) */
    Registry::make_resident(*att->gen);
    if (settings.refViewNr < att->gen->decode_error.size() && att->gen->decode_error[settings.refViewNr])
        std::rethrow_exception(att->gen->decode_error[settings.refViewNr]);      /* loadColorImage(scale), dmrecon.cc:79 */
    Slot* slot = att->slot;
    int32_t w = 0, h = 0;
    {
        std::lock_guard<std::mutex> lock(slot->parent_mu);
        if (mi_dmrecon_level_size(slot->parent, (int32_t)settings.refViewNr, settings.scale, &w, &h) != 0)
            throw std::invalid_argument("Invalid master view");
    }
    this->width = w;
    this->height = h;
    trace("DMRecon constructed (view)", t_ctor, (long)settings.refViewNr);
    if (!settings.quiet)
        std::cout << "scaled image size: " << this->width << " x " << this->height << std::endl;
}

void
DMRecon::start()
{
    progress.start_time = std::time(nullptr);
    if (progress.cancelled) { progress.status = RECON_CANCELLED; return; }
    mi_dmrecon_settings st;
    mi_dmrecon_settings_default(&st);
    st.filterWidth = (int32_t)settings.filterWidth;
    st.minNCC = settings.minNCC; st.minParallax = settings.minParallax;
    st.acceptNCC = settings.acceptNCC; st.minRefineDiff = settings.minRefineDiff;
    st.maxIterations = (int32_t)settings.maxIterations;
    st.nrReconNeighbors = (int32_t)settings.nrReconNeighbors;
    st.globalVSMax = (int32_t)settings.globalVSMax;
    st.scale = settings.scale;
    st.useColorScale = settings.useColorScale ? 1 : 0;
    for (int k = 0; k < 3; ++k) { st.aabbMin[k] = settings.aabbMin[k]; st.aabbMax[k] = settings.aabbMax[k]; }

    mve::FloatImage::Ptr depthImg = mve::FloatImage::create(width, height, 1);
    mve::FloatImage::Ptr dzImg = mve::FloatImage::create(width, height, 2);
    mve::FloatImage::Ptr confImg = mve::FloatImage::create(width, height, 1);
    mi_dmrecon_maps maps;
    maps.depth = depthImg->get_data_pointer();
    maps.dz = dzImg->get_data_pointer();
    maps.conf = confImg->get_data_pointer();
    maps.normal = nullptr;
    maps.views = nullptr;

    /* The library polls `cancelled` and publishes status/filled/queueSize through its own POD. */
    mi_dmrecon_progress mp;
    mp.status = MI_RECON_IDLE; mp.filled = 0; mp.queueSize = 0; mp.start_time = progress.start_time; mp.cancelled = 0;
    std::atomic<bool> running(true);
    std::thread relay([&]() {
        while (running.load()) {
            if (progress.cancelled) mp.cancelled = 1;
            progress.status = (ReconStatus)mp.status;
            progress.filled = mp.filled;
            progress.queueSize = mp.queueSize;
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    int32_t ref = (int32_t)settings.refViewNr;
    Slot* sl = std::static_pointer_cast<Attachment>(this->slot)->slot;
    Request req;
    req.st = st; req.ref = ref; req.maps = maps; req.prog = &mp;
    double const t_submit = trace_ms();
    /* filed first, withdrawn second: a gathering thread that sees the count reach zero finds every request */
    sl->file(req);
    std::static_pointer_cast<Attachment>(this->slot)->withdraw();
    sl->submit(req);                            /* returns when the batch this view ended up in has finished */
    trace("start(): request served (view)", t_submit, (long)ref);
    double const t_save = trace_ms();
    int rc = req.rc;
    running.store(false);
    relay.join();
    if (rc == MI_DMRECON_ECANCELLED) { progress.status = RECON_CANCELLED; return; }
    if (rc == MI_DMRECON_ENOIMAGE) {
        /* the global view selection picked a view whose image could not be decoded: the reference fails here with the decoder's
         * own exception (it loads the selected views, dmrecon.cc:236-240) -- the first such view in the order of the selection */
        progress.status = RECON_IDLE;
        Generation const& gen = *std::static_pointer_cast<Attachment>(this->slot)->gen;
        int32_t ids[MI_DMRECON_MAX_GLOBAL_VIEWS]; int32_t n_ids = 0;
        {
            std::lock_guard<std::mutex> lock(sl->parent_mu);
            if (mi_dmrecon_global_view_selection(sl->parent, &st, ref, ids, &n_ids) != 0) n_ids = 0;
        }
        for (int32_t k = 0; k < n_ids; ++k)
            if (ids[k] >= 0 && (std::size_t)ids[k] < gen.decode_error.size() && gen.decode_error[ids[k]])
                std::rethrow_exception(gen.decode_error[ids[k]]);
        throw std::runtime_error(req.err.empty() ? "a selected neighbour view has no image" : req.err);
    }
    if (rc != 0) {
        progress.status = RECON_IDLE;
        switch (rc) {
            case MI_DMRECON_EINVAL: throw std::invalid_argument(req.err);
            case MI_DMRECON_EFOOTPRINT: throw std::out_of_range(req.err);
            default: throw std::runtime_error(req.err);
        }
    }

    progress.status = RECON_SAVING;
    {
        std::size_t filled = 0;                 /* per view (the library's counters are per call = per batch) */
        float const* d = depthImg->get_data_pointer();
        for (int i = 0; i < width * height; ++i) filled += d[i] > 0.0f ? 1 : 0;
        progress.filled = filled;
    }
    mve::View::Ptr view = scene->get_views()[settings.refViewNr];
    if (settings.writePlyFile) {
        /* SingleView::saveReconAsPly (single_view.cc:122-138) through MVE's own exporters: the triangulated
         * depth map with confidences and the colours of the scaled image, plus the .xf camera file */
        std::string fname("mvs-");                                   /* SingleView::createFileName, single_view.h:144-151 */
        fname += util::string::get_filled(view->get_id(), 4);
        fname += "-L";
        fname += util::string::get((float)settings.scale);
        if (!settings.quiet)
            std::cout << "Saving ply file as " << settings.plyPath << "/" << fname << ".ply" << std::endl;
        if (settings.plyPath.empty()) throw std::invalid_argument("Empty path");
        if (!util::fs::dir_exists(settings.plyPath.c_str())) util::fs::mkdir(settings.plyPath.c_str());
        mve::ByteImage::Ptr color = mve::ByteImage::create(width, height, 3);
        {
            std::lock_guard<std::mutex> lock(sl->parent_mu);
            rc = mi_dmrecon_get_level(sl->parent, ref, settings.scale, color->get_data_pointer(), nullptr, nullptr);
        }
        if (rc != 0) raise_from(rc);
        mve::geom::save_ply_view(util::fs::join_path(settings.plyPath, fname + ".ply"), view->get_camera(),
            depthImg, confImg, color);
        mve::geom::save_xf_file(util::fs::join_path(settings.plyPath, fname + ".xf"), view->get_camera());
    }
    std::string name("depth-L");
    name += util::string::get(settings.scale);
    view->set_image(depthImg, name);
    if (settings.keepDzMap) {
        name = "dz-L";
        name += util::string::get(settings.scale);
        view->set_image(dzImg, name);
    }
    if (settings.keepConfidenceMap) {
        name = "conf-L";
        name += util::string::get(settings.scale);
        view->set_image(confImg, name);
    }
    if (settings.scale != 0) {
        mve::ByteImage::Ptr undist = mve::ByteImage::create(width, height, 3);
        {
            std::lock_guard<std::mutex> lock(sl->parent_mu);
            rc = mi_dmrecon_get_level(sl->parent, ref, settings.scale, undist->get_data_pointer(), nullptr, nullptr);
        }
        if (rc != 0) raise_from(rc);
        name = "undist-L";
        name += util::string::get(settings.scale);
        view->set_image(undist, name);
    }
    progress.status = RECON_IDLE;
    trace("start(): images set (view)", t_save, (long)ref);
    if (!settings.quiet) {
        float percent = (float)progress.filled / (float)(width * height);
        std::cout << "Filled " << progress.filled << " pixels, i.e. "
                  << util::string::get_fixed(percent * 100.f, 1) << " %." << std::endl;
        std::cout << "MVS took " << (std::time(nullptr) - progress.start_time) << " seconds." << std::endl;
    }
}

MVS_NAMESPACE_END
