// multi_gpu.inl -- several GPUs behind ONE scene handle (included by api.cpp, inside its unnamed namespace).
//
// The reference's fan-out is one call: Camera.createImage() scatters the pixels over availableProcessors() threads
// (Camera.java:129-182).  Here the one call is mi355rt_render / mi355rt_render_device on a scene that was given a device
// list (mi355rt_scene_set_devices): the scene's arrays are replicated on every listed GPU (device-to-device copies of the
// built arrays: nothing is rebuilt), rank r = the r-th device renders the frame's tiles t with t % N == r into ONE
// tile-compact block (the kernels' settings.tile_compact form: nothing to pack), the blocks travel to rank 0's device in
// the frame's ONE exchange step, and one kernel there puts them in image order.
//
//   host threads   one per device: the caller's thread serves rank 0, a persistent worker serves each other rank (a
//                  frame is four to a dozen launches per rank: issued from one thread, eight ranks' launches would take
//                  longer than an eighth of the frame does)
//   streams        one per rank and handle, on the rank's device; every rank's stream first waits for the caller's
//                  stream (an event), the caller's stream at the end waits for rank 0's: the call is asynchronous on
//                  `hip_stream` exactly as the one-GPU call is
//   exchange       RCCL (librccl.so.1, loaded on first use): one communicator per device from ncclCommInitAll, rank r
//                  ncclSend()s its block, rank 0 posts the N - 1 ncclRecv()s as one group -- point-to-point over xGMI,
//                  every peer on its own link into the root.  Or peer copies (hipMemcpyPeerAsync from the rank's stream
//                  + an event rank 0's stream waits for): taken when the list names a device twice (a rehearsal on one
//                  GPU: RCCL refuses duplicate devices), when librccl is not there, or when the switch
//                  MI355RT_TRANSPORT=peer says so.
//   frames in flight  a view of such a scene (mi355rt_scene_view) gets views of every replica, its own streams and
//                  blocks, and shares the workers and communicators; frames are issued one at a time (RCCL wants every
//                  rank's calls in the same order) and run side by side on the GPUs.

constexpr int TRANSPORT_NONE = 0, TRANSPORT_PEER = 1, TRANSPORT_RCCL = 2;

// librccl.so is half a gigabyte: it is loaded when a scene first asks for several distinct GPUs, never linked
struct RcclApi {
   void *lib = nullptr;
   ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
   ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
   ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*GroupStart)() = nullptr;
   ncclResult_t (*GroupEnd)() = nullptr;
   const char *(*GetErrorString)(ncclResult_t) = nullptr;
   bool ok = false;
};

const RcclApi &rccl_api()
{
   static const RcclApi api = [] {
      RcclApi a;
      for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
         a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
         if (a.lib) break;
      }
      if (!a.lib) return a;
      a.CommInitAll = (decltype(a.CommInitAll)) dlsym(a.lib, "ncclCommInitAll");
      a.CommDestroy = (decltype(a.CommDestroy)) dlsym(a.lib, "ncclCommDestroy");
      a.Send = (decltype(a.Send)) dlsym(a.lib, "ncclSend");
      a.Recv = (decltype(a.Recv)) dlsym(a.lib, "ncclRecv");
      a.GroupStart = (decltype(a.GroupStart)) dlsym(a.lib, "ncclGroupStart");
      a.GroupEnd = (decltype(a.GroupEnd)) dlsym(a.lib, "ncclGroupEnd");
      a.GetErrorString = (decltype(a.GetErrorString)) dlsym(a.lib, "ncclGetErrorString");
      a.ok = a.CommInitAll && a.CommDestroy && a.Send && a.Recv && a.GroupStart && a.GroupEnd && a.GetErrorString;
      return a;
   }();
   return api;
}

int rccl_fail(ncclResult_t e, const char *what)
{
   const RcclApi &api = rccl_api();
   return fail(MI355RT_E_DEVICE, std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(e) : "RCCL error"));
}
#define RCCL_TRY(expr)                                   \
   do {                                                  \
      ncclResult_t e_ = (expr);                          \
      if (e_ != ncclSuccess) return rccl_fail(e_, #expr); \
   } while (0)

// One persistent host thread per rank >= 1, bound to the rank's device.  post() hands it one job; wait() returns the
// job's status and error text (jobs report through the thread-local g_error like every entry point).
class Worker {
 public:
   explicit Worker(int device) : device_(device), thread_([this] { loop(); }) {}
   ~Worker()
   {
      {
         std::lock_guard<std::mutex> lock(m_);
         stop_ = true;
      }
      cv_.notify_all();
      thread_.join();
   }
   void post(std::function<int()> job)
   {
      {
         std::lock_guard<std::mutex> lock(m_);
         job_ = std::move(job);
         pending_ = true;
         done_ = false;
      }
      cv_.notify_all();
   }
   int wait(std::string &err)
   {
      std::unique_lock<std::mutex> lock(m_);
      cv_.wait(lock, [this] { return done_; });
      err = err_;
      return rc_;
   }

 private:
   void loop()
   {
      const bool device_ok = hipSetDevice(device_) == hipSuccess;
      for (;;) {
         std::function<int()> job;
         {
            std::unique_lock<std::mutex> lock(m_);
            cv_.wait(lock, [this] { return pending_ || stop_; });
            if (stop_) return;
            job = std::move(job_);
            pending_ = false;
         }
         int rc;
         std::string err;
         if (!device_ok) {
            rc = MI355RT_E_DEVICE;
            err = "cannot select HIP device " + std::to_string(device_);
         } else {
            rc = guarded(job);
            if (rc != MI355RT_OK) err = g_error;
         }
         {
            std::lock_guard<std::mutex> lock(m_);
            rc_ = rc;
            err_ = err;
            done_ = true;
         }
         cv_.notify_all();
      }
   }
   int device_;
   std::mutex m_;
   std::condition_variable cv_;
   std::function<int()> job_;
   bool pending_ = false, done_ = true, stop_ = false;
   int rc_ = MI355RT_OK;
   std::string err_;
   std::thread thread_;   // last: started when everything above exists
};

// what the handles of one replicated scene share: the device list, the exchange's transport and communicators, the workers
struct MultiShared {
   std::vector<int> devices;
   int transport = TRANSPORT_PEER;
   std::vector<ncclComm_t> comms;                  // TRANSPORT_RCCL: one per rank
   std::vector<std::unique_ptr<Worker>> workers;   // [r - 1] serves rank r
   std::mutex issue;                               // one frame's launches at a time, in the same order on every rank
   int refs = 0;                                   // handles (the scene and its views) that point here
};

struct RankCtx {
   int device = 0;
   mi355rt_scene *s = nullptr;   // rank 0: the handle itself; else the replica on `device` (a view of it, for a view)
   hipStream_t stream = nullptr;
   hipEvent_t sent = nullptr;    // peer copies: the rank's block has arrived on rank 0's device
   hipEvent_t t_begin = nullptr, t_end = nullptr;   // around the rank's kernels of the most recent frame
   bool timed = false;
   float *block = nullptr;       // 6 n words on the rank's device (rank 0 renders into the receive area itself)
   size_t block_words = 0;
};

struct Multi {
   MultiShared *shared = nullptr;
   std::vector<RankCtx> ranks;
   float *recv = nullptr;        // rank 0's device: the N blocks side by side, what the unpack kernel reads
   size_t recv_words = 0;
   hipEvent_t started = nullptr, unpacked = nullptr;
   bool unpacked_recorded = false;
};

void release_shared(MultiShared *sh)
{
   if (!sh || --sh->refs > 0) return;
   sh->workers.clear();   // joins the threads
   if (!sh->comms.empty()) {
      const RcclApi &api = rccl_api();
      for (ncclComm_t c : sh->comms)
         if (c && api.CommDestroy) (void) api.CommDestroy(c);
   }
   delete sh;
}

// frees what `s` holds on the other GPUs (and the multi-GPU state on its own); s->device is current
void release_multi(mi355rt_scene *s)
{
   Multi *m = s->multi;
   if (!m) return;
   s->multi = nullptr;
   for (size_t r = 0; r < m->ranks.size(); r++) {
      RankCtx &rk = m->ranks[r];
      DeviceScope scope(rk.device);
      if (rk.stream) {
         (void) hipStreamSynchronize(rk.stream);
         (void) hipStreamDestroy(rk.stream);
      }
      if (rk.sent) (void) hipEventDestroy(rk.sent);
      if (rk.t_begin) (void) hipEventDestroy(rk.t_begin);
      if (rk.t_end) (void) hipEventDestroy(rk.t_end);
      if (rk.block) (void) hipFree(rk.block);
      if (r > 0 && rk.s) release_scene_impl(rk.s);   // the replica (or the view of it)
   }
   if (m->recv) (void) hipFree(m->recv);
   if (m->started) (void) hipEventDestroy(m->started);
   if (m->unpacked) (void) hipEventDestroy(m->unpacked);
   release_shared(m->shared);
   delete m;
}

// A copy of the scene's read-only arrays on another device.  Everything in them is an index, never an address
// (layout.h: links are absolute within the arrays), so the bytes are valid wherever they lie.
int replicate_scene(const mi355rt_scene *root, int device, mi355rt_scene **out)
{
   *out = nullptr;
   DeviceScope scope(device);
   if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device " + std::to_string(device));
   mi355rt_scene *c = new mi355rt_scene(*root);   // host-side description: objects, materials, bounds, flags, sizes
   c->device = device;
   c->parent = nullptr;
   c->n_views = 0;
   c->dying = false;
   c->multi = nullptr;
   c->d_nodes = c->d_descs = c->d_gfaces = c->d_fnrm = nullptr;
   c->d_slot_ids = nullptr;
   c->d_objs = nullptr;
   c->d_ext = nullptr;
   c->d_brdf_dirs = c->d_brdf_weights = nullptr;
   c->d_counters = nullptr;
   c->h_readback = nullptr;
   c->noalloc_scratch = nullptr;
   c->scratch = FrameScratch();
   c->pending_times.clear();
   c->event_pool.clear();
   c->d_rgb = nullptr;
   c->d_argb8 = nullptr;
   c->d_hit_id = c->d_hit_obj = nullptr;
   c->d_hit_w = nullptr;
   c->frame_pixels = 0;
   c->copy_stream = nullptr;
   c->device_bytes = 0;
   c->stats_valid = false;
   struct Guard {
      mi355rt_scene *s;
      ~Guard() { release_scene_impl(s); }
   } guard { c };
   struct Arr {
      void **dst;
      const void *src;
      size_t bytes;
   };
   const size_t n_obj = std::max<size_t>(root->objects.size(), 1);
   const Arr arrs[] = {
      { &c->d_nodes, root->d_nodes, std::max<size_t>(root->n_node_recs, 1) * sizeof(DevNodeRec) },
      { &c->d_descs, root->d_descs, (root->n_descs + DESC_PAD) * sizeof(LeafDesc) },
      { &c->d_gfaces, root->d_gfaces, root->n_gface_floats * sizeof(float) },
      { (void **) &c->d_slot_ids, root->d_slot_ids, std::max<size_t>(root->n_slots, 1) * sizeof(int32_t) },
      { &c->d_fnrm, root->d_fnrm, std::max<size_t>(root->n_fnrm_floats * sizeof(float), 64) },
      { (void **) &c->d_objs, root->d_objs, n_obj * sizeof(DevObject) },
      { (void **) &c->d_ext, root->d_ext, n_obj * sizeof(DevObjectExt) },
      { (void **) &c->d_brdf_dirs, root->d_brdf_dirs, root->d_brdf_dirs ? 3 * (size_t) root->n_dirs * sizeof(float) : 0 },
      { (void **) &c->d_brdf_weights, root->d_brdf_weights, root->d_brdf_weights ? (size_t) root->n_dirs * (size_t) root->n_materials * sizeof(float) : 0 },
   };
   for (const Arr &a : arrs) {
      if (!a.src || !a.bytes) continue;
      HIP_TRY(hipMalloc(a.dst, a.bytes));
      c->device_bytes += a.bytes;
      HIP_TRY(hipMemcpyPeer(*a.dst, device, a.src, root->device, a.bytes));
   }
   HIP_TRY(hipMalloc((void **) &c->d_counters, CNT_WORDS * sizeof(uint32_t)));
   HIP_TRY(hipHostMalloc((void **) &c->h_readback, N_STATS * sizeof(unsigned long long), hipHostMallocDefault));
   if (noalloc_scratch_ptr(device, &c->noalloc_scratch) != hipSuccess) {
      c->noalloc_scratch = nullptr;
      (void) hipGetLastError();
   }
   int cus = 0;
   if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = cus;
   guard.s = nullptr;
   *out = c;
   return MI355RT_OK;
}

int make_view(mi355rt_scene *root, mi355rt_scene **view_out);   // api.cpp, below: one device's view

// streams, events of every rank of a handle whose ranks[].s and device are filled in; s->device is current
int init_multi_handles(Multi *m)
{
   for (RankCtx &rk : m->ranks) {
      DeviceScope scope(rk.device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device " + std::to_string(rk.device));
      HIP_TRY(hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&rk.sent, hipEventDisableTiming));
      HIP_TRY(hipEventCreate(&rk.t_begin));
      HIP_TRY(hipEventCreate(&rk.t_end));
   }
   HIP_TRY(hipEventCreateWithFlags(&m->started, hipEventDisableTiming));
   HIP_TRY(hipEventCreateWithFlags(&m->unpacked, hipEventDisableTiming));
   return MI355RT_OK;
}

// mi355rt_scene_set_devices
int set_devices_impl(mi355rt_scene *s, const int32_t *devices, int32_t n)
{
   if (s->parent) return fail(MI355RT_E_ARG, "give the device list to the scene, not to a view of it");
   if (s->n_views > 0) return fail(MI355RT_E_ARG, "give the device list before creating views of the scene");
   if (s->multi) return fail(MI355RT_E_ARG, "the scene has a device list already");
   if (n < 1 || n > MI355RT_MAX_RANKS) return fail(MI355RT_E_ARG, "the device list holds 1 to " + std::to_string(MI355RT_MAX_RANKS) + " devices");
   int n_dev = 0;
   if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(MI355RT_E_DEVICE, "no HIP device available");
   std::vector<int> devs(devices, devices + n);
   if (devs[0] < 0) devs[0] = s->device;
   if (devs[0] != s->device) return fail(MI355RT_E_ARG, "the first device of the list must be the scene's own (" + std::to_string(s->device) + "): the frame is gathered there");
   bool distinct = true;
   for (int r = 0; r < n; r++) {
      if (devs[r] < 0 || devs[r] >= n_dev) return fail(MI355RT_E_ARG, "device ordinal " + std::to_string(devs[r]) + " out of range");
      for (int q = 0; q < r; q++) distinct = distinct && devs[q] != devs[r];
   }
   const Switches sw = switches();
   std::unique_ptr<MultiShared> sh(new MultiShared());
   sh->devices = devs;
   sh->transport = TRANSPORT_PEER;
   if (distinct && sw.transport != TRANSPORT_PEER && (n > 1 || sw.transport == TRANSPORT_RCCL)) {
      const RcclApi &api = rccl_api();
      if (api.ok) {
         sh->comms.assign((size_t) n, nullptr);
         const ncclResult_t e = api.CommInitAll(sh->comms.data(), n, devs.data());
         if (e == ncclSuccess) sh->transport = TRANSPORT_RCCL;
         else sh->comms.clear();
         if (e != ncclSuccess && sw.transport == TRANSPORT_RCCL) return rccl_fail(e, "ncclCommInitAll");
      } else if (sw.transport == TRANSPORT_RCCL) {
         return fail(MI355RT_E_DEVICE, "MI355RT_TRANSPORT=rccl, but librccl.so.1 could not be loaded");
      }
   } else if (sw.transport == TRANSPORT_RCCL) {
      return fail(MI355RT_E_ARG, "MI355RT_TRANSPORT=rccl needs a list of distinct devices (RCCL refuses a device named twice)");
   }
   (void) hipGetLastError();
   for (int r = 1; r < n; r++) {   // peer access both ways where the hardware has it (copies fall back to staging otherwise)
      if (devs[r] == devs[0]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devs[0], devs[r]) == hipSuccess && can) (void) hipDeviceEnablePeerAccess(devs[r], 0);
      DeviceScope scope(devs[r]);
      if (scope.ok() && hipDeviceCanAccessPeer(&can, devs[r], devs[0]) == hipSuccess && can) (void) hipDeviceEnablePeerAccess(devs[0], 0);
      (void) hipGetLastError();   // "already enabled" is not an error here
   }
   std::unique_ptr<Multi> m(new Multi());
   m->shared = sh.get();
   m->ranks.resize((size_t) n);
   m->ranks[0].device = devs[0];
   m->ranks[0].s = s;
   s->multi = m.get();   // from here on release_multi(s) undoes everything
   sh->refs = 1;
   sh.release();
   Multi *mp = m.release();
   for (int r = 1; r < n; r++) {
      mp->ranks[r].device = devs[r];
      if (int rc = replicate_scene(s, devs[r], &mp->ranks[r].s)) {
         release_multi(s);
         return rc;
      }
   }
   if (int rc = init_multi_handles(mp)) {
      release_multi(s);
      return rc;
   }
   for (int r = 1; r < n; r++) mp->shared->workers.emplace_back(new Worker(devs[r]));
   return MI355RT_OK;
}

// a view of a scene with a device list: views of every replica, own streams and blocks, shared workers and communicators
int view_multi(mi355rt_scene *root, mi355rt_scene *view)
{
   const Multi *rm = root->multi;
   std::unique_ptr<Multi> m(new Multi());
   m->shared = rm->shared;
   m->ranks.resize(rm->ranks.size());
   m->ranks[0].device = rm->ranks[0].device;
   m->ranks[0].s = view;
   view->multi = m.get();
   rm->shared->refs++;
   Multi *mp = m.release();
   for (size_t r = 1; r < rm->ranks.size(); r++) {
      mp->ranks[r].device = rm->ranks[r].device;
      DeviceScope scope(mp->ranks[r].device);
      if (int rc = make_view(rm->ranks[r].s, &mp->ranks[r].s)) {
         release_multi(view);
         return rc;
      }
   }
   if (int rc = init_multi_handles(mp)) {
      release_multi(view);
      return rc;
   }
   return MI355RT_OK;
}

// one rank's share of a frame, on the rank's own thread with the rank's device current
int rank_frame(Multi *m, int r, const mi355rt_camera *cam, const mi355rt_settings *st_in, int tile, size_t n, bool self_send)
{
   RankCtx &rk = m->ranks[(size_t) r];
   const int world = (int) m->ranks.size();
   const bool into_recv = r == 0 && !self_send;
   float *block = into_recv ? m->recv : rk.block;   // (sized by render_multi before anything was posted)
   mi355rt_settings st = *st_in;
   st.tile_size = tile;
   st.tile_rank = r;
   st.tile_world = world;
   st.tile_compact = 1;
   FrameArgs p;
   long long n_micro = 0;
   // (what this checks is the same on every rank -- camera, settings, the scene's description -- so either every rank goes
   // on to the exchange or none does)
   if (int rc = fill_frame_args(rk.s, cam, &st, p, n_micro)) return rc;
   p.rgb = block;
   p.argb8 = nullptr;
   p.hit_id = (int32_t *) (block + 3 * n);
   p.hit_obj = (int32_t *) (block + 4 * n);
   p.hit_w = block + 5 * n;
   // The frame itself can fail on ONE rank (its scratch grows on a first, larger frame: device memory).  The exchange below
   // is posted all the same: a send that never came would leave rank 0's receives, and a receive that never came the other
   // ranks' sends, waiting for good (RCCL has no time-outs).  The call then fails with the rank's error.
   const int frame_rc = [&]() -> int {
      HIP_TRY(hipStreamWaitEvent(rk.stream, m->started, 0));                               // after what the caller's stream held
      if (m->unpacked_recorded) HIP_TRY(hipStreamWaitEvent(rk.stream, m->unpacked, 0));    // the receive area is free again
      HIP_TRY(hipEventRecord(rk.t_begin, rk.stream));
      if (int rc = run_frame(rk.s, p, n_micro, (st.collect_stats & MI355RT_COLLECT_TIMES) != 0, rk.stream)) return rc;
      HIP_TRY(hipEventRecord(rk.t_end, rk.stream));
      rk.timed = true;
      return MI355RT_OK;
   }();
   const std::string frame_err = frame_rc != MI355RT_OK ? g_error : std::string();
   const int exchange_rc = [&]() -> int {
   MultiShared *sh = m->shared;
   if (sh->transport == TRANSPORT_RCCL) {
      const RcclApi &api = rccl_api();
      if (r == 0) {   // the N - 1 receives as one group (and, in the one-rank rehearsal, the send to itself with them)
         RCCL_TRY(api.GroupStart());
         ncclResult_t e = ncclSuccess;
         if (self_send) e = api.Send(block, 6 * n, ncclFloat, 0, sh->comms[0], rk.stream);
         if (self_send && e == ncclSuccess) e = api.Recv(m->recv, 6 * n, ncclFloat, 0, sh->comms[0], rk.stream);
         for (int q = 1; q < world && e == ncclSuccess; q++) e = api.Recv(m->recv + (size_t) q * 6 * n, 6 * n, ncclFloat, q, sh->comms[0], rk.stream);
         const ncclResult_t e2 = api.GroupEnd();
         if (e != ncclSuccess) return rccl_fail(e, "ncclRecv");
         if (e2 != ncclSuccess) return rccl_fail(e2, "ncclGroupEnd");
      } else {
         RCCL_TRY(api.Send(block, 6 * n, ncclFloat, 0, sh->comms[(size_t) r], rk.stream));
      }
   } else if (r > 0) {
      HIP_TRY(hipMemcpyPeerAsync(m->recv + (size_t) r * 6 * n, m->ranks[0].device, block, rk.device, 6 * n * sizeof(float), rk.stream));
      HIP_TRY(hipEventRecord(rk.sent, rk.stream));
   }
   return MI355RT_OK;
   }();
   if (frame_rc != MI355RT_OK) return fail(frame_rc, frame_err);
   return exchange_rc;
}

// mi355rt_render_device on a handle with a device list; s->device is current
int render_multi(mi355rt_scene *s, const mi355rt_camera *cam, const mi355rt_settings *st, const mi355rt_frame *frame, hipStream_t stream)
{
   Multi *m = s->multi;
   MultiShared *sh = m->shared;
   if (!cam) return fail(MI355RT_E_ARG, "camera is null");
   if (int rc = check_settings(st)) return rc;
   if (st->tile_world > 1 || st->tile_compact) return fail(MI355RT_E_ARG, "a scene with a device list shards the frame itself: tile_world / tile_compact must be unset");
   if (cam->width < 1 || cam->height < 1) return fail(MI355RT_E_ARG, "image size must be positive");
   const int world = (int) m->ranks.size();
   const int tile = st->tile_size == 0 ? 16 : st->tile_size;
   if (tile < 8 || tile % 8 != 0 || tile > 4096) return fail(MI355RT_E_ARG, "tile_size must be a multiple of 8 in [8, 4096]");
   const long long tiles = (long long) ((cam->width + tile - 1) / tile) * ((cam->height + tile - 1) / tile);
   const long long slots = (tiles + world - 1) / world;
   const size_t n = (size_t) slots * (size_t) tile * (size_t) tile;   // pixels of a block
   const bool self_send = sh->transport == TRANSPORT_RCCL && world == 1;
   std::lock_guard<std::mutex> lock(sh->issue);
   if (m->recv_words < (size_t) world * 6 * n) {
      if (m->recv) {
         HIP_TRY(hipDeviceSynchronize());
         for (RankCtx &rk : m->ranks) {   // a rank's copy into the old area may still be on its way
            DeviceScope scope(rk.device);
            HIP_TRY(hipStreamSynchronize(rk.stream));
         }
         (void) hipFree(m->recv);
      }
      m->recv = nullptr;
      m->recv_words = 0;
      HIP_TRY(hipMalloc((void **) &m->recv, (size_t) world * 6 * n * sizeof(float)));
      m->recv_words = (size_t) world * 6 * n;
      m->unpacked_recorded = false;
   }
   for (int r = self_send ? 0 : 1; r < world; r++) {   // every rank's block, before anything is posted (see rank_frame)
      RankCtx &rk = m->ranks[(size_t) r];
      if (rk.block_words >= 6 * n) continue;
      DeviceScope scope(rk.device);
      if (!scope.ok()) return fail(MI355RT_E_DEVICE, "cannot select HIP device " + std::to_string(rk.device));
      if (rk.block) {
         HIP_TRY(hipStreamSynchronize(rk.stream));
         (void) hipFree(rk.block);
      }
      rk.block = nullptr;
      rk.block_words = 0;
      HIP_TRY(hipMalloc((void **) &rk.block, 6 * n * sizeof(float)));
      rk.block_words = 6 * n;
   }
   HIP_TRY(hipEventRecord(m->started, stream));
   for (int r = 1; r < world; r++) sh->workers[(size_t) r - 1]->post([=]() -> int { return rank_frame(m, r, cam, st, tile, n, false); });
   int rc = rank_frame(m, 0, cam, st, tile, n, self_send);
   std::string err = rc != MI355RT_OK ? g_error : std::string();
   for (int r = 1; r < world; r++) {   // every worker is waited for, whatever happened: the jobs hold pointers to this call's arguments
      std::string werr;
      const int wrc = sh->workers[(size_t) r - 1]->wait(werr);
      if (wrc != MI355RT_OK && rc == MI355RT_OK) {
         rc = wrc;
         err = "rank " + std::to_string(r) + " (device " + std::to_string(m->ranks[(size_t) r].device) + "): " + werr;
      }
   }
   if (rc != MI355RT_OK) return fail(rc, err);
   hipStream_t s0 = m->ranks[0].stream;
   if (sh->transport == TRANSPORT_PEER)
      for (int r = 1; r < world; r++) HIP_TRY(hipStreamWaitEvent(s0, m->ranks[(size_t) r].sent, 0));
   HIP_TRY(launch_tiles_unpack(m->recv, world, (int) slots, tile, cam->width, cam->height, frame->rgb, frame->argb8, frame->hit_id, frame->hit_obj,
                               frame->hit_w, s0));
   HIP_TRY(hipEventRecord(m->unpacked, s0));
   m->unpacked_recorded = true;
   HIP_TRY(hipStreamWaitEvent(stream, m->unpacked, 0));
   return MI355RT_OK;
}
