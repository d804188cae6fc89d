// One device-resident decode call: the launch sequence (k_size -> k_scan+k_layout -> k_init -> k_emit -> k_publish),
// its settlement, and the in-call split over internal streams.
#include "engine_internal.h"

using namespace rhe;

// One device-resident decode call.  enqueue() puts the whole call on the stream (k_size -> k_scan+k_layout -> k_init ->
// k_emit -> one D2H of the control words) and finish() waits for it and settles the result (error check, arena
// retry, host tables).  rh_decode_device runs both back to back; with RH_ASYNC the result is handed out between the
// two and rh_device_result_wait() (or the first accessor that needs a host-side fact) runs finish(): the caller's next
// call is on the stream before this one has drained, which is what a pipeline of small batches needs -- a 1M-record
// call is 0.15 ms of kernels behind ~25 us of host turn-around (profiles/r03q_timeline_*.txt).
struct rh_decode_call {
  // the call
  rh_schema* s;
  const CompiledSchema& cs;
  const uint8_t* d_data;
  const uint64_t* d_offsets;
  uint64_t data_len, n, num_chunks;
  rh_opts opts;                 // by value: an asynchronous call outlives the caller's struct
  bool want_stats;
  rh_stats st;
  rh_device_result& r;
  HostProf hp;
  int device = 0;
  hipStream_t stream = nullptr;
  ChunkGeo geo_v;
  const ChunkGeo* geo = nullptr;
  // derived
  uint32_t k = 1;
  int K = 0, nnodes = 0, nbuf = 0;
  const DeviceProgram* dp = nullptr;
  const SpecKernel* sk = nullptr;
  hipFunction_t size_r = nullptr, emit_r = nullptr;     // the ranged pair, when this call launches it (tiles past the LDS window)
  uint64_t narrow_rows = 0, tile = 0, bpc64 = 0, payload = 0;
  uint32_t nblocks = 0;
  uint64_t o_null = 0, o_tot = 48, ctrl_bytes = 0;
  uint32_t null_slots = rh::kNullSlots;      // program.h null_slots_for(k)
  Lease ws, hctrl, dtab, prof_buf;
  std::unique_ptr<CtrlLease> ctrl;
  rh::KParams P;
  uint32_t lds_bytes = 0, emit_lds = 0;
  bool profile = false;
  Events ev;
  std::vector<uint64_t> totals;
  uint64_t n_entries = 0, tab_bytes = 0;
  uint64_t* d_sizes = nullptr;
  uint64_t exact = 0;
  bool child_bitmaps = false, fused = false, timed_size = false;
  bool range_of_host_call = false;   // a chunk range of a host call (decode_range) or a group of a split call: the caller gave the geometry
  bool single = false;          // the single-pass form ran (rh_spec_fused): arena laid out from capacities
  std::vector<uint64_t> caps;   // [K][k] those capacities
  uint64_t arena_cap = 0, o_tick = 0;
  Lease lookback, hcaps;
  double basis = 0;
  bool settled = false;         // finish() ran (or the call completed inside enqueue())
  bool async = false;           // RH_ASYNC: the call is settled later; without rh_k_publish its end is marked with a DoneEvent
  DoneEvent done;
  // in-call overlap (decode_device_split, staggered form): this group's size pass starts behind `start_after` (the previous
  // group's size pass) and marks its own end with `sized`, so that size pass g+1 runs beside emit pass g
  hipEvent_t start_after = nullptr, sized = nullptr;
  bool published = false;       // rh_k_publish ran: hctrl holds the compact layout (summed null counts) behind a token
  uint32_t token = 0;
  uint64_t o_flag_h = 0;

  rh_decode_call(rh_schema* s_, const uint8_t* data, const uint64_t* offs, uint64_t dl, uint64_t n_, uint64_t nc, const rh_opts* o,
               bool stats, const ChunkGeo* g, rh_device_result& res)
      : s(s_), cs(*s_->cs), d_data(data), d_offsets(offs), data_len(dl), n(n_), num_chunks(nc), opts(o ? *o : default_opts()),
        want_stats(stats), r(res) {
    std::memset(&st, 0, sizeof st);
    if (g) { geo_v = *g; geo = &geo_v; range_of_host_call = true; }
    opts.devices = nullptr; opts.n_devices = 0; opts.device_stats = nullptr; opts.ready = nullptr; opts.gathered = nullptr;   // (not used below; never dangling)
  }

  // Where the call's Arrow buffers go: HBM, or -- host calls, RH_INTERNAL_HOST_ARENA -- a pooled block of PINNED HOST memory that
  // the emit kernel fills through the PCIe link and that is then lent to the result (engine_export.cpp to_host_impl): no D2H
  // copy, no host wait between kernels and copy, and the copy engine is left to the H2D copies of the next chunk group (the two
  // directions on SDMA shared ~56 GB/s between them: a 1M-record call's D2H copies ran at 32 GB/s beside its H2D copies,
  // profiles/r05h_*).  The CUs' stores reach the full rate of the link: 53 GB/s, what a D2H copy of the same bytes reaches
  // (profiles/r05i_host_arena.txt).  Only with a block that sits idle in the pool -- allocating one costs 0.2 ms per MB, so the
  // pool is refilled in the background (Pool::prefetch) -- and not for schemas whose child-domain bitmaps are built with
  // atomics on the arena.  RUHVRO_HIP_HOST_ARENA=0 turns it off (A/B).
  Lease arena_lease(uint64_t bytes) {
    r.arena_host = false;
    if ((opts.flags & RH_INTERNAL_HOST_ARENA) && !child_bitmaps && env_long("RUHVRO_HIP_HOST_ARENA", 1, 0, 1) != 0) {
      Block b = pin_pool().try_get(bytes, device);
      if (b.p) {
        r.arena_host = true;
        return Lease(pin_pool(), b);
      }
      pin_pool().prefetch(bytes + bytes / 16, device, pinned_budget_left());
    }
    return Lease(dev_pool(), bytes, device);
  }

  void check_bad(const uint8_t* h) {
    unsigned long long fb = *(const unsigned long long*)h;
    if (!fb) return;
    const uint64_t rec = ~fb;
    uint64_t c = r.sz ? std::min<uint64_t>(rec / r.sz, k - 1) : 0;
    uint64_t b = c * bpc64 + (rec - c * r.sz) / tile;
    rh::ErrInfo ei;
    HIPCHK(hipMemcpy(&ei, P.errinfo + b, sizeof ei, hipMemcpyDeviceToHost));
    throw DecodeError(format_error(ei));
  }

  // host statement of the layout (same rule, same table order as rh_k_layout): fills the result's tables
  void layout_host() {
    r.data_bytes = totals;
    for (auto t : totals)
      if (t > 0x7FFFFFFFull) {
        count(RH_CTR_OFFSET32_ERRORS);
        throw DecodeError("offset overflow: a chunk's column exceeds the 2^31-1 limit of 32-bit Arrow offsets");
      }
    if (sk)
      for (int d = 1; d < cs.ndom; d++)
        for (uint32_t c = 0; c < k; c++)
          if (totals[(size_t)(d - 1) * k + c] >= narrow_rows) throw NeedWideIndex();
    r.fill_tables();
    exact = r.output_bytes;
  }

  void launch_tail(bool offsets_done) {     // k_init + k_emit through the device tables at dtab
    if (nbuf > 0 && (child_bitmaps || !offsets_done) &&
        rh_launch_init(P.bufptr, d_sizes, dp->desc, (uint32_t)nbuf, k, P.first_bad, stream)) throw HipError("k_init launch failed");
    if (n > 0) {
      emit_lds = lds_bytes;
      if (sk ? launch_module(sk->emit_fn, P, nblocks, (uint32_t)tile, emit_lds, stream, ev.at(3), P.ranged ? nullptr : ev.at(4))
             : rh_launch_emit(&P, emit_lds, stream, ev.at(3), ev.at(4)))
        throw HipError("k_emit launch failed");
      if (P.ranged && launch_module(emit_r, P, nblocks + (P.worklist ? rh::kBigFront : 0u), (uint32_t)tile, emit_lds, stream, nullptr, ev.at(4))) throw HipError("k_emit (ranged) launch failed");
    } else {
      ev.rec(3, stream);
      ev.rec(4, stream);
    }
  }

  void exact_tail() {      // totals are on the host: exactly sized arena, tables from the host
    ctrl->b.clean = false;
    published = false;       // (the raw device layout is copied back below)
    layout_host();
    r.arena = arena_lease(r.arena_bytes);
    Lease htab(pin_pool(), tab_bytes, device);
    void** hptr = (void**)htab.ptr();
    uint64_t* hsz = (uint64_t*)(htab.ptr() + (uint64_t)std::max(nbuf, 1) * k * 8);
    for (uint32_t c = 0; c < k; c++)
      for (int b = 0; b < nbuf; b++) {   // device tables are [chunk][buf]
        hptr[(size_t)c * nbuf + b] = r.arena.ptr() + r.buf_off[(size_t)b * k + c];
        hsz[(size_t)c * nbuf + b] = r.buf_size[(size_t)b * k + c];
      }
    HIPCHK(hipMemcpyAsync(dtab.ptr(), htab.ptr(), tab_bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(ctrl->ptr() + 8, 0, 8, stream));    // clear the layout flag (and the ticket) of a refused optimistic attempt
    launch_tail(false);
    // (not the totals: the host has them, and rh_k_publish may have zeroed the device copy of a refused attempt)
    HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), o_tot, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(hctrl.ptr() + o_null, ctrl->ptr() + o_null, ctrl_bytes - o_null, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));      // also keeps htab alive until the table copy is done
    check_bad(hctrl.ptr());
  }

  // The single-pass form (spec_body.h spec_fused): k_layout over per-column CAPACITIES from the schema's history, then ONE
  // kernel that sizes, scans across tiles (look-back) and emits, then rh_k_publish.  Returns false when the call does not
  // qualify (no history yet, generic kernels, knobs) -- the two-pass submission follows then.
  bool try_single(bool two_sync, long ratio_hook) {
    const bool on = (opts.flags & RH_SINGLE_PASS) != 0 || env_long("RUHVRO_HIP_SINGLE_PASS", kSinglePassDefault, 0, 1) != 0;
    // (device-resident calls only: a host call is bound by the PCIe link, and its D2H copy would carry the capacity slack)
    if (!on || range_of_host_call || (opts.flags & (RH_INTERNAL_TWO_PASS | RH_TWO_PASS)) || !sk || K <= 0 || K > 64 || n == 0 || two_sync ||
        ratio_hook >= 0)
      return false;
    if (n_entries > (1u << 16)) return false;
    if (8ull * K * k > 4ull * K * nblocks) return false;      // the capacities travel in the workspace's blocksum area (below)
    // the single-pass kernel is its own code object, compiled when a call first asks for it (in the background unless the
    // caller insists on specialised kernels): until it is there the call takes the two-pass form
    hipFunction_t fused_fn = sk->fused_fn.load(std::memory_order_acquire);
    if (!fused_fn) {
      if (sk->fused_dead) return false;
      fused_fn = spec_kernel(s, device, compile_policy(opts.flags & 3, n), false, true).fused_fn.load(std::memory_order_acquire);
      if (!fused_fn) return false;
    }
    std::vector<double> per_row;
    {
      std::lock_guard<std::mutex> g(s->mu);
      if ((int)s->per_row.size() != K) return false;
      if (s->single_cooldown > 0) { s->single_cooldown--; return false; }
      per_row = s->per_row;
    }
    // capacities: what the last call needed per row, + 4 % + a pad that covers a short chunk's noise; never more than the
    // 32-bit limits the kernels index with (a column that needs more overflows its capacity -> two-pass -> the usual errors)
    // (RUHVRO_HIP_SINGLE_SLACK_PERMILLE: knob / test hook -- below 1000 the capacities are smaller than what the last call
    //  needed, which forces the LF_CAPACITY fail-over to the two-pass form)
    const double slack = (double)env_long("RUHVRO_HIP_SINGLE_SLACK_PERMILLE", 1040, 1, 4000) / 1000.0;
    caps.assign((size_t)K * k, 0);
    for (int kk = 0; kk < K; kk++)
      for (uint32_t c = 0; c < k; c++) {
        const uint64_t rows_c = c == k - 1 ? r.rows_last : r.sz;
        uint64_t cap = (uint64_t)(per_row[(size_t)kk] * (double)rows_c * slack) + (slack >= 1.0 ? 4096 : 0);
        uint64_t lim = 0x7FFFFFFFull;
        if (kk < cs.ndom - 1) lim = std::min<uint64_t>(lim, narrow_rows - 1);       // a child row domain
        caps[(size_t)kk * k + c] = std::min(cap, lim);
      }
    {   // arena bytes of that layout (the rule of fill_tables / rh_k_layout)
      uint64_t off = 0;
      for (uint32_t c = 0; c < k; c++)
        for (int b = 0; b < nbuf; b++) {
          const rh::BufDesc& d = cs.bufs[b];
          const uint64_t rows0 = c == k - 1 ? r.rows_last : r.sz;
          const uint64_t rows = d.dom == 0 ? rows0 : caps[(size_t)(d.dom - 1) * k + c];
          off += rh::buf_slot_bytes(rh::buf_bytes(d.kind, rows, d.kind == rh::BK_DATA ? caps[(size_t)d.counter * k + c] : 0, nullptr, (uint32_t)d.counter));
        }
      arena_cap = std::max<uint64_t>(off, kAlign);
    }
    count(RH_CTR_SINGLE_PASS_CALLS);
    count(RH_CTR_FUSED_CALLS);             // (a single stream submission too)
    single = true; fused = true;
    r.arena = Lease(dev_pool(), arena_cap, device);
    lookback = Lease(dev_pool(), std::max<uint64_t>(8ull * K * nblocks, kAlign), device);
    HIPCHK(hipMemsetAsync(lookback.ptr(), 0, 8ull * K * nblocks, stream));
    // the capacities go to the device behind the leading words of the workspace's blocksum area (unused on this path)
    hcaps = Lease(pin_pool(), 8ull * K * k, device);
    std::memcpy(hcaps.ptr(), caps.data(), 8ull * K * k);
    uint64_t* d_caps = (uint64_t*)P.blocksum;
    HIPCHK(hipMemcpyAsync(d_caps, hcaps.ptr(), 8ull * K * k, hipMemcpyHostToDevice, stream));
    P.lookback = (unsigned long long*)lookback.ptr();
    P.caps = d_caps;
    rh::LParams LP;
    std::memset(&LP, 0, sizeof LP);
    LP.totals = d_caps; LP.desc = dp->desc; LP.sz = r.sz; LP.rows_last = r.rows_last; LP.n = n; LP.k = k;
    LP.nbuf = nbuf; LP.K = K; LP.ndom = cs.ndom; LP.arena = r.arena.ptr(); LP.capacity = r.arena.b.size;
    LP.bufptr = (void**)dtab.ptr(); LP.bufsize = d_sizes; LP.ctrl = P.first_bad; LP.narrow = 0;
    LP.narrow_rows = narrow_rows;
    if (rh_launch_layout(&LP, stream)) throw HipError("k_layout launch failed");
    if (nbuf > 0 && child_bitmaps && rh_launch_init(P.bufptr, d_sizes, dp->desc, (uint32_t)nbuf, k, P.first_bad, stream))
      throw HipError("k_init launch failed");
    const uint64_t tiles_max = std::max<uint64_t>((r.sz + tile - 1) / tile, (r.rows_last + tile - 1) / tile);
    emit_lds = lds_bytes;
    if (launch_module(fused_fn, P, (uint32_t)(tiles_max * k), (uint32_t)tile, emit_lds, stream, ev.at(3), ev.at(4)))
      throw HipError("k_fused launch failed");
    basis = (double)payload + 64.0 * (double)n;
    void* hdev = nullptr;
    if (hipHostGetDevicePointer(&hdev, hctrl.ptr(), 0) == hipSuccess && hdev) {
      static std::atomic<uint32_t> next_token{0x40000001u};
      token = next_token.fetch_add(1);
      if (token == 0) token = next_token.fetch_add(1);
      o_flag_h = align_up(o_null + 4ull * nnodes * k, 8);
      *(volatile uint32_t*)(hctrl.ptr() + o_flag_h) = 0;
      if (rh_launch_publish(ctrl->ptr(), hdev, (uint32_t)(o_null / 4), (uint32_t)(nnodes * (int)k), (uint32_t)(o_flag_h / 4), token, null_slots, P.tileflag, 0u /* no size pass: nobody wrote the tile flags */, 8u, stream))
        throw HipError("k_publish launch failed");
      ctrl->b.clean = true;
      published = true;
    } else {
      (void)hipGetLastError();
      HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
      if (async) done.record(device, stream);
    }
    return true;
  }

  void enqueue() {
    Range rk("ruhvro_hip:decode_device (k_size, k_scan, k_layout, k_init, k_emit)");
    if (opts.device >= 0) { HIPCHK(hipSetDevice(opts.device)); device = opts.device; }
    else HIPCHK(hipGetDevice(&device));
    stream = (hipStream_t)opts.stream;
    if ((uintptr_t)d_data & 15) throw std::invalid_argument("device payload pointer must be 16-byte aligned");

    {   // the generic kernels' dynamic-LDS limit is a per-device function attribute: set it once per device
      static std::mutex lds_mu;
      static std::vector<int> lds_done;
      std::lock_guard<std::mutex> g(lds_mu);
      if (std::find(lds_done.begin(), lds_done.end(), device) == lds_done.end()) {
        if (rh_set_max_lds(160 * 1024) != 0) throw HipError("cannot raise the dynamic LDS limit of the decode kernels");
        lds_done.push_back(device);
      }
    }

    r.cs = &cs;
    r.device = device;
    r.n = n;
    if (!geo && opts.chunk_rows) {   // a range of a larger call's chunks (one process per GPU): rh_opts.chunk_rows
      if (num_chunks < 1 || num_chunks > 0xFFFFFFFFull || (num_chunks - 1) > n / opts.chunk_rows ||
          (n > 0 && n == (num_chunks - 1) * opts.chunk_rows && num_chunks > 1))
        throw std::invalid_argument("chunk_rows: the n records do not make num_chunks chunks of chunk_rows rows (the last one takes the rest)");
      geo_v.k = (uint32_t)num_chunks;
      geo_v.sz = opts.chunk_rows;
      geo_v.rows_last = n - (num_chunks - 1) * opts.chunk_rows;
      geo_v.payload_bytes = data_len;
      geo = &geo_v;
    }
    k = geo ? geo->k : rh_clamp_chunks(n, num_chunks);
    r.k = k;
    r.sz = geo ? geo->sz : n / k;
    r.rows_last = geo ? geo->rows_last : n - (uint64_t)(k - 1) * r.sz;
    K = cs.K; nnodes = (int)cs.nodes.size(); nbuf = (int)cs.bufs.size();
    dp = &device_program(s, device);

    // kernel form: schema-specialised (compiled once per schema, cached) or the generic interpreter
    const int mode = opts.flags & 3;
    // the specialised kernels address every chunk buffer with 32-bit byte offsets
    // (every chunk buffer below 4 GiB: at most max_row_bytes per row -- 16 unless the schema has a wider fixed)
    // (RUHVRO_HIP_NARROW_ROWS: test hook that lowers the bound so that small inputs take the wide-index fallback)
    narrow_rows = (uint64_t)env_long("RUHVRO_HIP_NARROW_ROWS",
                                     (long)std::min<uint64_t>(1ull << 28, (1ull << 32) / std::max<uint32_t>(cs.max_row_bytes, 16)), 1, 1l << 28);
    const bool narrow_ok = std::max(r.sz, r.rows_last) < narrow_rows;
    if (mode != RH_KERNEL_GENERIC && n > 0 && narrow_ok) {
      const SpecKernel& k0 = spec_kernel(s, device, compile_policy(mode, n));
      if (k0.ok) sk = &k0;
      else if (mode == RH_KERNEL_SPECIALIZED) throw HipError("specialised kernel unavailable: " + k0.why);
    }
    // The ranged pair behind the size / emit kernels (spec_body.h ranged_tile): while the schema's recent calls met tiles past the
    // LDS window (rh_schema::ranged_calls, fed by rh_k_publish's tile statistics), or when the caller insists on the specialised
    // kernels.  A call that meets such tiles without it takes the fallback (the careful walk from global memory) and, large enough,
    // starts the pair's compile in the background.  RUHVRO_HIP_RANGED=0 / 1: never / always (A/B, tests).
    if (sk) {
      const long force = env_long("RUHVRO_HIP_RANGED", -1, 0, 1);
      const bool want = force == 1 || (force != 0 && (s->ranged_calls.load(std::memory_order_relaxed) > 0 || mode == RH_KERNEL_SPECIALIZED));
      if (want && !sk->ranged_dead) {
        hipFunction_t er = sk->emit_r_fn.load(std::memory_order_acquire);
        // (asked for by the schema's history: never wait for the compile -- the generic kernels take the call meanwhile)
        if (!er) er = spec_kernel(s, device, (force == 1 || mode == RH_KERNEL_SPECIALIZED) ? rh::CP_BLOCKING : rh::CP_BACKGROUND, false, false, true).emit_r_fn.load(std::memory_order_acquire);
        if (er) { emit_r = er; size_r = sk->size_r_fn.load(std::memory_order_acquire); }
      }
      if (want && !emit_r) {
        if (mode == RH_KERNEL_SPECIALIZED) throw HipError("specialised kernels for tiles past the LDS window unavailable");
        sk = nullptr;            // tiles past the window are expected and the pair is not there (yet): the generic kernels
      }
    }
    // records per workgroup: a wide schema's tile is one wavefront on both kernel forms (program.h kWideTile)
    tile = cs.wide ? (uint64_t)rh::kWideTile : sk ? (uint64_t)rh::spec_tile_records() : (uint64_t)rh::kBlock;
    bpc64 = std::max<uint64_t>((r.sz + tile - 1) / tile, 1);
    const uint64_t nblocks64 = n == 0 ? 0 : (uint64_t)(k - 1) * bpc64 + (r.rows_last + tile - 1) / tile;
    if (nblocks64 > 0x7FFFFFFFull / std::max(K, 1)) throw std::invalid_argument("too many records for one call");
    nblocks = (uint32_t)nblocks64;

    // ---- control block: [first_bad u64 | layout flag, ticket | arena bytes | pad][totals u64 K*k][nullcount u32 nnodes*k*null_slots]
    //      workspace: errinfo | blocksum | blockbase | tileflag | lanecnt
    o_tot = 48;      // control words first (program.h): first_bad, layout flag, arena bytes used, pad; words 8..11 = the tile statistics rh_k_publish sums
    o_tick = o_tot + 8ull * K * k;                    // [k] tile tickets of the single-pass form (zero like the rest of the block)
    o_null = align_up(o_tick + 4ull * k, 16);
    null_slots = rh::null_slots_for(k);
    // (with one null-count slot -- k > 2048 chunks -- the slot area is no larger than the compact host layout, whose publish
    //  token sits behind the summed counts: room for it, whatever the rounding)
    ctrl_bytes = align_up(std::max<uint64_t>(o_null + 4ull * nnodes * k * null_slots, align_up(o_null + 4ull * nnodes * k, 8) + 8), kAlign);
    const uint64_t o_err = 0;       // the rest lives in the workspace (needs no zeroing)
    const uint64_t o_bsum = align_up(o_err + sizeof(rh::ErrInfo) * (uint64_t)nblocks, kAlign);
    const uint64_t o_bbase = align_up(o_bsum + 4ull * K * nblocks, kAlign);
    const uint64_t o_flag = align_up(o_bbase + 4ull * K * nblocks, kAlign);
    const uint64_t o_lcnt = align_up(o_flag + 4ull * nblocks, kAlign);
    // (per-record counters from the size pass to the emit pass: two per dword for the specialised kernels, one for the generic ones)
    const uint64_t o_lcnt32 = align_up(o_lcnt + 4ull * (sk ? (uint64_t)((cs.KL + 1) / 2) : (uint64_t)cs.KL) * nblocks * tile, kAlign);
    const uint64_t o_wl = align_up(o_lcnt32 + ((size_r && emit_r) ? 4ull * (uint64_t)cs.KL * nblocks * tile : 0), kAlign);      // (+ the ranged pair's 32-bit counters)
    const uint64_t ws_bytes = align_up(o_wl + ((size_r && emit_r) ? 8ull + 8ull * nblocks : 0), kAlign);                           // (+ its work list: large tiles first)
    hp.mark("setup");
    ws = Lease(dev_pool(), ws_bytes, device);
    hctrl = Lease(pin_pool(), ctrl_bytes, device);
    ctrl.reset(new CtrlLease(ctrl_bytes, device, stream));        // all zero (CtrlPool)
    hp.mark("leases");

    std::memset(&P, 0, sizeof P);
    P.data = d_data; P.offsets = d_offsets; P.data_len = data_len;
    P.n = n; P.sz = r.sz; P.rows_last = r.rows_last; P.k = k; P.bpc = (uint32_t)bpc64; P.nblocks = nblocks;
    P.prog = dp->prog; P.sym_off = dp->sym_off; P.sym_data = dp->sym_data;
    P.nops = (int)cs.prog.size(); P.K = K; P.KL = cs.KL; P.tile = (uint32_t)tile; P.ndom = cs.ndom; P.nnodes = nnodes; P.list_depth = cs.list_depth;
    P.nbuf = nbuf; P.cnt_databuf = dp->cnt_databuf;
    P.ranged = (size_r && emit_r) ? 1u : 0u;
    P.first_bad = (unsigned long long*)ctrl->ptr();
    P.nullcount = (uint32_t*)(ctrl->ptr() + o_null);
    P.null_slots = null_slots;
    P.totals = (uint64_t*)(ctrl->ptr() + o_tot);
    P.tickets = (uint32_t*)(ctrl->ptr() + o_tick);
    P.errinfo = (rh::ErrInfo*)(ws.ptr() + o_err);
    P.blocksum = (uint32_t*)(ws.ptr() + o_bsum);
    P.blockbase = (uint32_t*)(ws.ptr() + o_bbase);
    P.tileflag = (uint32_t*)(ws.ptr() + o_flag);
    P.lanecnt = (uint32_t*)(ws.ptr() + o_lcnt);
    P.lanecnt32 = (uint32_t*)(ws.ptr() + o_lcnt32);
    P.worklist = (P.ranged && K > 0 && n > 0) ? (uint32_t*)(ws.ptr() + o_wl) : nullptr;      // (filled by the size kernel: only with a size pass)
    P.bigmark = P.worklist ? P.worklist + 2 + nblocks : nullptr;

    // LDS: fixed part + input window sized from the mean record length (falls back to global reads
    // for workgroups whose 256 records do not fit)
    const uint32_t lds_fixed = (sk ? rh::spec_lds_fixed_words_host(K, nnodes, (int)(tile / 64), rh::child_bitmap_count(cs), rh::dense_list_count(cs), rh::dom0_bitmap_count(cs)) * 4 : rh_lds_fixed_bytes(K, cs.KL, (int)tile, cs.list_depth, nnodes, nbuf)) + 16;   // + window slack
    payload = geo ? geo->payload_bytes : data_len;
    const uint64_t avg = n ? payload / n + 1 : 16;
    // (tuning / test knobs, read per call: RUHVRO_HIP_WIN_PCT, RUHVRO_HIP_WIN_PAD)
    const uint64_t win_pct = (uint64_t)env_long("RUHVRO_HIP_WIN_PCT", 115, 100, 400);
    const uint64_t win_pad = (uint64_t)env_long("RUHVRO_HIP_WIN_PAD", 2048, 0, 65536);
    uint64_t win = align_up(avg * tile * win_pct / 100 + win_pad * tile / rh::kBlock, 16);
    win = std::max<uint64_t>(win, 8192 * tile / rh::kBlock);
    const uint64_t lds_cap = 160 * 1024 - 512;
    if (lds_fixed + 4096 > lds_cap) throw rh::SchemaError("schema needs more LDS than a CDNA4 workgroup has");
    win = std::min<uint64_t>(win, std::min<uint64_t>((lds_cap - lds_fixed) & ~15ull, 96 * 1024));
    // Occupancy steps: a CU's 160 KB hold N workgroups of at most 160 KB / N each.  A window that puts the workgroup just
    // above a step costs a whole workgroup per CU (a quarter of the resident waves at N = 4) for a few hundred bytes of
    // slack, so it gives that slack up as long as a smaller margin (6 % + 1 KB over the mean tile) is left.
    if (win_pct == 115 && win_pad == 2048) {       // (not when a test / sweep sets the window by hand)
      const uint64_t min_win = align_up(avg * tile * 106 / 100 + 1024 * tile / rh::kBlock, 16);
      for (uint64_t nwg = 4; nwg >= 2; nwg--) {        // (4: what the emit kernel's registers allow at most)
        const uint64_t step = (160 * 1024 / nwg) & ~511ull;
        if (lds_fixed + win > step && step > lds_fixed && step - lds_fixed >= min_win) { win = (step - lds_fixed) & ~15ull; break; }
      }
      // Round 6: never a window that leaves a CU fewer than four workgroups.  Records too large for it (a mean tile of 40 KB and
      // more: ~150 B per record) are walked in ranges by the ranged kernels, which keeps sixteen wavefronts per CU resident --
      // measured on the skewed workload (mean record 393 B): 7.1 ms with a 40 KB window, 10.7 ms with the 100 KB one the mean
      // tile asks for (profiles/r06_c_*).  Only with the specialised kernels: the generic ones have no ranges.
      if (sk && !cs.wide) {
        const uint64_t step4 = (160 * 1024 / 4) & ~511ull;
        if (step4 > lds_fixed + 8192 && lds_fixed + win > step4) win = (step4 - lds_fixed) & ~15ull;
      }
    }
    // A wide schema's tile is one wavefront: the window is kept small enough for 8 of them per CU (records of a hundred and more
    // columns are large: a window for 64 of them would leave a CU one or two wavefronts) -- tiles past it are walked in ranges.
    if (cs.wide && win_pct == 115 && win_pad == 2048) {
      const uint64_t per_wave = (160 * 1024 / 8) & ~511ull;
      if (per_wave > lds_fixed + 4096) win = std::min<uint64_t>(win, (per_wave - lds_fixed) & ~15ull);
    }
    // (RUHVRO_HIP_WIN_BYTES: the window as given -- 0 = none, every tile is walked from global memory; measurement knob)
    {
      const long fixed_win = env_long("RUHVRO_HIP_WIN_BYTES", -1, 0, 150 * 1024);
      if (fixed_win >= 0) win = std::min<uint64_t>((uint64_t)fixed_win & ~15ull, (lds_cap - lds_fixed) & ~15ull);
    }
    P.win_bytes = (uint32_t)win;
    // (the ranged pair takes LARGE tiles first, program.h KParams::worklist: large = well beyond both the window and this call's mean
    //  tile -- every tile of a wide schema is several windows)
    P.big_tile_bytes = (uint64_t)rh::kBigTileWindows * std::max<uint64_t>(win, nblocks ? payload / nblocks : 0);
    lds_bytes = lds_fixed + (uint32_t)win;
    // optional in-kernel phase timing of the specialised kernels (RUHVRO_HIP_PROFILE=1)
    static const bool profile_env = [] { const char* e = std::getenv("RUHVRO_HIP_PROFILE"); return e && *e && *e != '0'; }();
    profile = profile_env;
    if (profile && sk) {
      prof_buf = Lease(dev_pool(), 64 * 32 * 8, device);
      HIPCHK(hipMemsetAsync(prof_buf.ptr(), 0, 64 * 32 * 8, stream));
      P.prof = (unsigned long long*)prof_buf.ptr();
    }

    if (want_stats) ev.init();
    hp.mark("events");

    // ---- the launch sequence.  With a size history for this schema the whole call is ONE stream submission:
    //   k_size -> k_scan -> k_layout (exact arena layout on the device, program.h LParams) -> k_init -> k_emit -> one D2H
    // of the control words.  The arena is reserved up front from the history; when it turns out too small (the data
    // changed character), the layout kernel says so, init/emit return at once, and the host re-runs the tail with an
    // exactly sized arena -- which is also what the first call of a schema does.
    totals.assign((size_t)K * k, 0);
    n_entries = (uint64_t)k * std::max(nbuf, 0);
    tab_bytes = align_up((uint64_t)std::max(nbuf, 1) * k * 16, kAlign);
    dtab = Lease(dev_pool(), tab_bytes, device);
    d_sizes = (uint64_t*)(dtab.ptr() + (uint64_t)std::max(nbuf, 1) * k * 8);
    P.bufptr = (void* const*)dtab.ptr();
    for (const rh::BufDesc& d : cs.bufs) child_bitmaps = child_bitmaps || (d.kind == rh::BK_BITMAP && d.dom != 0);   // built with atomics on zeroed words

    const bool two_sync = env_long("RUHVRO_HIP_TWO_SYNC", 0, 0, 1) != 0;
    // (RUHVRO_HIP_ARENA_PERMILLE: test hook, the arena is reserved as if the schema's history said that many output
    //  bytes per 1000 input bytes -- a small value forces the LF_CAPACITY retry)
    const long ratio_hook = env_long("RUHVRO_HIP_ARENA_PERMILLE", -1, 0, 1000000);
    const double ratio = ratio_hook >= 0 ? std::max(1e-9, ratio_hook / 1000.0) : s->arena_ratio.load();
    fused = n > 0 && ratio > 0 && !two_sync && n_entries <= (1u << 16);
    // stage timings (rh_stats) come from the kernels' own start / stop timestamps: e0..e1 = k_size, e5..e2 = k_scan,
    // e3..e4 = k_emit
    if (start_after) HIPCHK(hipStreamWaitEvent(stream, start_after, 0));
    if (try_single(two_sync, ratio_hook)) return;
    timed_size = n > 0 && K > 0;
    if (timed_size) {
      if (P.worklist) HIPCHK(hipMemsetAsync(P.worklist, 0, 8ull + 8ull * nblocks, stream));      // the list's length, every tile's mark

      // (with the ranged pair: the size kernel's start and the ranged size kernel's stop bracket the pass)
      if (sk ? launch_module(sk->size_fn, P, nblocks, (uint32_t)tile, lds_bytes, stream, ev.at(0), P.ranged ? nullptr : ev.at(1))
             : rh_launch_size(&P, lds_bytes, stream, ev.at(0), ev.at(1)))
        throw HipError("k_size launch failed");
      if (P.ranged && launch_module(size_r, P, nblocks + (P.worklist ? rh::kBigFront : 0u), (uint32_t)tile, lds_bytes, stream, nullptr, ev.at(1))) throw HipError("k_size (ranged) launch failed");
      if (sized) HIPCHK(hipEventRecord(sized, stream));
      // (the single-submission path scans and lays the arena out in ONE launch, below)
      if (!fused && rh_launch_scan(&P, stream, ev.at(5), ev.at(2))) throw HipError("k_scan launch failed");
    } else {
      // no size pass (no variable-length output): nobody classified the tiles, so the emit kernel walks all of them carefully
      P.all_careful = 1;
      if (sized) HIPCHK(hipEventRecord(sized, stream));
    }
    // RUHVRO_HIP_NO_TRUST=1 (debugging aid): the emit pass walks EVERY tile with its own bounds and anomaly checks instead of
    // trusting the size pass's verdict on the same bytes (walk.h RH_TRUST) -- what a caller that suspects its input buffers
    // change between the two passes of an RH_ASYNC call turns on; the GPU suite passes with it (tests/test_async_device.py)
    static const bool no_trust = env_long("RUHVRO_HIP_NO_TRUST", 0, 0, 1) != 0;
    if (no_trust) P.all_careful = 1;
    hp.mark("size+scan_launch");
    basis = (double)payload + 64.0 * (double)n;
    if (fused) {
      count(RH_CTR_FUSED_CALLS);
      const uint64_t capacity = align_up((uint64_t)(ratio * basis * 1.125) + n_entries * kAlign + (1u << 20), kAlign);
      r.arena = arena_lease(capacity);
      rh::LParams LP;
      std::memset(&LP, 0, sizeof LP);
      LP.totals = P.totals; LP.desc = dp->desc; LP.sz = r.sz; LP.rows_last = r.rows_last; LP.n = n; LP.k = k;
      LP.nbuf = nbuf; LP.K = K; LP.ndom = cs.ndom; LP.arena = r.arena.ptr(); LP.capacity = r.arena.b.size;
      LP.bufptr = (void**)dtab.ptr(); LP.bufsize = d_sizes; LP.ctrl = P.first_bad; LP.narrow = sk ? 1u : 0u;
      LP.narrow_rows = narrow_rows;
      if (timed_size ? rh_launch_scan_layout(&P, &LP, stream, ev.at(5), ev.at(2)) : rh_launch_layout(&LP, stream))
        throw HipError("k_scan / k_layout launch failed");
      launch_tail(true);                           // the layout kernel wrote offsets[0] = 0 itself
      hp.mark("layout+emit_launch");
      // the control words go to the host from the call's last kernel, which also re-zeroes the block (rh_k_publish) and
      // writes a per-call token behind them: finish() spins on that word instead of waiting for a stream event
      void* hdev = nullptr;
      static const bool no_publish = env_long("RUHVRO_HIP_NO_PUBLISH", 0, 0, 1) != 0;
      if (!no_publish && hipHostGetDevicePointer(&hdev, hctrl.ptr(), 0) == hipSuccess && hdev) {
        static std::atomic<uint32_t> next_token{1};
        token = next_token.fetch_add(1);
        if (token == 0) token = next_token.fetch_add(1);
        o_flag_h = align_up(o_null + 4ull * nnodes * k, 8);                 // host layout: head | compact null counts | token
        *(volatile uint32_t*)(hctrl.ptr() + o_flag_h) = 0;
        if (rh_launch_publish(ctrl->ptr(), hdev, (uint32_t)(o_null / 4), (uint32_t)(nnodes * (int)k), (uint32_t)(o_flag_h / 4), token, null_slots, P.tileflag, (K > 0 && n > 0) ? nblocks : 0u, 8u, stream))
          throw HipError("k_publish launch failed");
        ctrl->b.clean = true;
        published = true;
      } else {
        (void)hipGetLastError();
        HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
        if (async) done.record(device, stream);
      }
      hp.mark("d2h_enqueue");
    }
  }

  void finish() {
    if (settled) return;
    settled = true;
    HIPCHK(hipSetDevice(device));
    if (fused) {
      if (published) {
        // spin on the token rh_k_publish stores last into this call's pinned block: this call only (later calls stay
        // queued behind it), no event in the stream, and sooner than a stream wait returns
        volatile uint32_t* flag = (volatile uint32_t*)(hctrl.ptr() + o_flag_h);
        for (uint32_t spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != token;) {
          if ((++spins & 0xFFFFu) == 0) {              // a failed launch or a fault must not hang the caller
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {                     // everything on the stream is done: the token must be there
              if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != token) throw HipError("rh_k_publish finished without publishing its token");
              break;
            }
            if (q != hipErrorNotReady) throw HipError(std::string("stream failed while waiting for a decode call: ") + hipGetErrorString(q));
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      } else if (done.e) {
        done.wait();                               // this call only: later calls stay queued behind it
      } else {
        HIPCHK(hipStreamSynchronize(stream));
      }
      hp.mark("sync");
      check_bad(hctrl.ptr());
      if (published && K > 0 && n > 0 && !single) {      // the tile statistics rh_k_publish summed (include/ruhvro_hip.h RH_CTR_*_TILES)
        const uint32_t* stw = (const uint32_t*)(hctrl.ptr() + 32);
        count(RH_CTR_TILES, nblocks);
        count(RH_CTR_CAREFUL_TILES, stw[0]); count(RH_CTR_OVER_WINDOW_TILES, stw[1]);
        count(RH_CTR_REWALKED_WAVES, stw[2]); count(RH_CTR_SUBTILED_TILES, stw[3]);
        // tiles past the window: this schema's next calls launch the ranged pair (and compile it if need be)
        if (stw[1]) s->ranged_calls.store(kRangedKeep, std::memory_order_relaxed);
        else { uint32_t v = s->ranged_calls.load(std::memory_order_relaxed); if (v) s->ranged_calls.compare_exchange_weak(v, v - 1, std::memory_order_relaxed); }
      }
      if (K > 0) std::memcpy(totals.data(), hctrl.ptr() + o_tot, 8ull * K * k);
      const uint32_t lflag = *(const uint32_t*)(hctrl.ptr() + 8);
      if (lflag & rh::LF_NEED_RANGED) {      // a tile past the LDS window and no ranged pair in this call: nothing was emitted
        ctrl->b.clean = published;
        s->ranged_calls.store(kRangedKeep, std::memory_order_relaxed);
        r.arena.release();
        throw NeedRanged();
      }
      if (single) {
        ctrl->b.clean = published;
        if (lflag) {            // a column outgrew its capacity (or the capacity layout was refused): the two-pass path decides
          // (not latched when the capacities were shrunk by the test hook)
          if ((lflag & rh::LF_CAPACITY) && env_long("RUHVRO_HIP_SINGLE_SLACK_PERMILLE", 1040, 1, 4000) >= 1000) {
            std::lock_guard<std::mutex> g(s->mu);
            s->single_backoff = std::min<uint32_t>(1024, std::max<uint32_t>(8, s->single_backoff * 2));
            s->single_cooldown = s->single_backoff;
          }
          count(RH_CTR_SINGLE_PASS_FAILOVERS);
          r.arena.release();
          throw NeedTwoPass();
        }
        r.data_bytes = totals;
        r.layout_bytes = caps;
        r.arena_bytes = arena_cap;
        { std::lock_guard<std::mutex> g(s->mu); s->single_backoff = 0; }
      } else if (lflag & rh::LF_CAPACITY) {
        count(RH_CTR_CAPACITY_RETRIES);
        r.arena.release();
        exact_tail();                              // (throws the offset-overflow / wide-index cases itself)
      } else if (lflag || want_stats) {
        layout_host();                             // throws for LF_OFFSET32 / LF_NEED_WIDE: same tests on the same totals
        if (lflag) throw HipError("internal error: layout kernel and host disagree");
        if (r.arena_bytes != std::max<uint64_t>(*(const uint64_t*)(hctrl.ptr() + 16), kAlign)) throw HipError("internal error: device and host arena layouts differ");
      } else {
        // the device laid the arena out and accepted it: the host's tables (same rule, same totals) wait for their first
        // reader (rh_device_result::tables) -- a caller that only hands the device buffers on never pays for them
        r.data_bytes = totals;
        r.arena_bytes = std::max<uint64_t>(*(const uint64_t*)(hctrl.ptr() + 16), kAlign);
      }
    } else {
      count(RH_CTR_TWO_SYNC_CALLS);
      if (n > 0 && K > 0) {
        HIPCHK(hipMemcpyAsync(hctrl.ptr(), ctrl->ptr(), ctrl_bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        check_bad(hctrl.ptr());
        if (*(const uint32_t*)(hctrl.ptr() + 8) & rh::LF_NEED_RANGED) {
          ctrl->b.clean = false;
          s->ranged_calls.store(kRangedKeep, std::memory_order_relaxed);
          throw NeedRanged();
        }
        std::memcpy(totals.data(), hctrl.ptr() + o_tot, 8ull * K * k);
      }
      exact_tail();
    }
    if (n > 0 && basis > 0 && !single) {
      const double slots = (double)n_entries * (double)kAlign;
      s->arena_ratio.store(std::max(0.0, (double)r.arena_bytes - slots) / basis + 1e-9);
    }
    if (n > 0 && K > 0 && (int)totals.size() == K * (int)k) {       // per-row need of every counter's column (single-pass capacities)
      std::vector<double> pr((size_t)K, 0.0);
      for (int kk = 0; kk < K; kk++)
        for (uint32_t c = 0; c < k; c++) {
          const uint64_t rows_c = c == k - 1 ? r.rows_last : r.sz;
          if (rows_c) pr[(size_t)kk] = std::max(pr[(size_t)kk], (double)totals[(size_t)kk * k + c] / (double)rows_c);
        }
      std::lock_guard<std::mutex> g(s->mu);
      s->per_row = std::move(pr);
    }
    r.nullcount.assign((size_t)nnodes * k, 0);
    if (published) {           // rh_k_publish summed the slots: one word per (node, chunk)
      std::memcpy(r.nullcount.data(), hctrl.ptr() + o_null, 4ull * nnodes * k);
    } else {
      const uint32_t* slots = (const uint32_t*)(hctrl.ptr() + o_null);      // [nnodes][k][null_slots] (program.h)
      for (size_t e = 0; e < (size_t)nnodes * k; e++) {
        uint32_t sum = 0;
        for (uint32_t sl = 0; sl < null_slots; sl++) sum += slots[e * null_slots + sl];
        r.nullcount[e] = sum;
      }
    }
    hp.mark("host_layout");

    if (profile && sk) {
      unsigned long long hr[64 * 32], h[32] = {0};
      HIPCHK(hipMemcpy(hr, prof_buf.ptr(), sizeof hr, hipMemcpyDeviceToHost));
      for (int r0 = 0; r0 < 64; r0++)
        for (int i = 0; i < 32; i++) h[i] += hr[r0 * 32 + i];
      const double waves = (double)nblocks * 4;
      static const char* names2[] = {"offsets", "stage+barrier", "lane_init", "walk1", "scan", "barrier", "layout", "walk2",
                                     "errors+barrier", "flush"};
      static const char* names1[] = {"ticket+zero+barrier", "offsets+stage+barrier", "size_walk", "wave_scan", "barrier", "lookback(wave0)",
                                     "barrier", "prefix", "emit_walk", "errors+flush"};
      const char* const* names = single ? names1 : names2;
      std::fprintf(stderr, "[ruhvro_hip profile] %s cycles/wave:", single ? "single-pass" : "emit");
      for (int i = 0; i < 10; i++) std::fprintf(stderr, " %s=%.0f", names[i], h[i] / waves);
      std::fprintf(stderr, "\n[ruhvro_hip profile] size cycles/wave: stage+barrier=%.0f init=%.0f walk=%.0f tail=%.0f | kernels ms: size=%.3f emit=%.3f\n",
                   h[16] / waves, h[17] / waves, h[18] / waves, h[19] / waves, ev.ms(0, 1), ev.ms(3, 4));
    }
    if (want_stats) {
      st.records = n;
      st.input_bytes = payload;
      st.output_bytes = exact;
      st.chunks = k;
      st.blocks = nblocks;
      st.size_kernel_ms = (timed_size && !single) ? ev.ms(0, 1) : 0.f;
      st.scan_kernel_ms = (timed_size && !single) ? ev.ms(5, 2) : 0.f;
      st.emit_kernel_ms = n > 0 ? ev.ms(3, 4) : 0.f;
      st.specialized = sk ? 1 : 0;
      st.lds_bytes = emit_lds;
    }
    // the call's scratch goes back to the pools now (the control block is zeroed on its stream, CtrlPool)
    ws.release(); dtab.release(); hctrl.release(); prof_buf.release(); lookback.release(); hcaps.release(); ctrl.reset();
  }

  // a call that failed (or is abandoned) must not hand its blocks back while the GPU may still be using them
  void drain() noexcept {
    if (stream || device >= 0) { (void)hipSetDevice(device); (void)hipStreamSynchronize(stream); }
  }
};

rh_device_result::rh_device_result() { std::memset(&st, 0, sizeof st); }
rh_device_result::~rh_device_result() {
  if (pending) {                 // freed without a wait: the GPU may still be writing into the blocks this result owns
    pending->drain();
    pending.reset();
  }
  parts.clear();                 // (each group drains its own stream)
  if (!join_events.empty()) {
    std::lock_guard<std::mutex> g(DoneEvent::mu());
    auto& v = DoneEvent::idle()[device];
    for (hipEvent_t e : join_events) {
      if (v.size() < 64) v.push_back(e);
      else (void)hipEventDestroy(e);
    }
  }
}


namespace rhe {

typedef rh_decode_call DeviceDecode;

rh_device_result* decode_device_impl1(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len,
                                      uint64_t n, uint64_t num_chunks, const rh_opts* opts, rh_stats* stats, const ChunkGeo* geo) {
  auto res = std::make_unique<rh_device_result>();
  auto call = std::make_unique<DeviceDecode>(s, d_data, d_offsets, data_len, n, num_chunks, opts, stats != nullptr, geo, *res);
  const bool async = opts && (opts->flags & RH_ASYNC) && !geo;
  call->async = async;
  try {
    call->enqueue();
    if (async && call->fused) {            // everything is on the stream: settle later (rh_device_result_wait)
      res->pending = std::move(call);
      return res.release();
    }
    call->finish();
  } catch (...) {
    call->drain();
    throw;
  }
  if (stats) {
    const float pack = stats->pack_ms, h2d = stats->h2d_ms, d2h = stats->d2h_ms, tot = stats->total_ms;
    *stats = call->st;
    stats->pack_ms = pack; stats->h2d_ms = h2d; stats->d2h_ms = d2h; stats->total_ms = tot;
  }
  return res.release();
}


// ---------------------------------------------------------------------------
// In-call overlap: a large device-resident call deals its chunk GROUPS to internal streams.
//
// The reference runs one task per chunk (ruhvro/src/deserialize.rs:92-120); chunks are independent here too, and the two
// passes load different parts of a CU (the size pass is bound by VALU issue, the emit pass co-limited by the vector-memory
// path), so the size pass of one group running beside the emit pass of another fills issue slots that either kernel
// alone leaves empty (bench.py `overlapped` measured it between independent calls; this is the same inside ONE call).
// Group g = chunks [k*g/G, k*(g+1)/G) is a complete sub-call (size -> scan+layout -> emit -> publish, its own arena and
// control block) on its own stream; the caller's stream forks into the internal streams at the start of the call and
// joins them at its end, so the result is valid in stream order on rh_opts.stream exactly like an unsplit call's.
// Not taken when the caller asks for stage timings (kernels that share the chip have no per-kernel duration), on a
// schema's first call (no size history), for the generic kernels, or below RUHVRO_HIP_SPLIT_MIN records.
// RUHVRO_HIP_INTERNAL_STREAMS=G (1 = off) -- profiler passes run with 1.
// ---------------------------------------------------------------------------
void settle(rh_device_result* r);

hipEvent_t pooled_event(int device) {
  hipEvent_t e = nullptr;
  {
    std::lock_guard<std::mutex> g(DoneEvent::mu());
    auto& v = DoneEvent::idle()[device];
    if (!v.empty()) { e = v.back(); v.pop_back(); }
  }
  if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

// The internal streams of the in-call split on one device: created on first use, kept for the process, shared by every caller
// stream of that device (a split call forks into them from its own stream with an event and joins them back the same way, so
// they carry no caller's identity -- keyed by the caller's handle, as they were, a destroyed and re-created stream found stale
// companions and the 65th caller silently lost the split).  Calls that split at the same time on one device
// share them in stream order.
std::vector<hipStream_t> companion_streams(int device, hipStream_t /*caller*/, unsigned want) {
  static std::mutex mu;
  static std::map<int, std::vector<hipStream_t>> all;
  std::lock_guard<std::mutex> g(mu);
  auto& v = all[device];
  while (v.size() < want) {
    hipStream_t x = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    v.push_back(x);
  }
  return std::vector<hipStream_t>(v.begin(), v.begin() + want);
}

rh_device_result* decode_device_split(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len, uint64_t n,
                                      const rh_opts& opts, uint32_t k, uint64_t sz, uint64_t rows_last, unsigned NS) {
  int device = opts.device;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  hipStream_t caller = (hipStream_t)opts.stream;
  const std::vector<hipStream_t> extra = companion_streams(device, caller, NS - 1);
  if (extra.size() != NS - 1) return nullptr;
  // groups: G >= NS runs of whole chunks, dealt to the NS streams round-robin (RUHVRO_HIP_SPLIT_GROUPS, default = NS);
  // staggered (RUHVRO_HIP_SPLIT_STAGGER, default on): group g + 1's size pass starts when group g's has finished, so it
  // runs beside group g's EMIT pass (different bounds) instead of beside its size pass (the same bound)
  const unsigned G = (unsigned)std::min<uint64_t>(k, (uint64_t)std::max<long>(env_long("RUHVRO_HIP_SPLIT_GROUPS", 0, 0, 64), (long)NS));
  const bool stagger = env_long("RUHVRO_HIP_SPLIT_STAGGER", 1, 0, 1) != 0;
  hipEvent_t prev_sized = nullptr;
  auto parent = std::make_unique<rh_device_result>();
  parent->cs = s->cs.get(); parent->device = device; parent->n = n; parent->k = k; parent->sz = sz; parent->rows_last = rows_last;
  const bool async = (opts.flags & RH_ASYNC) != 0;
  count(RH_CTR_SPLIT_CALLS);
  // fork: the internal streams start behind everything that is on the caller's stream now (the input buffers' producers)
  hipEvent_t fork = pooled_event(device);
  parent->join_events.push_back(fork);
  HIPCHK(hipEventRecord(fork, caller));
  for (hipStream_t x : extra) HIPCHK(hipStreamWaitEvent(x, fork, 0));
  for (unsigned g = 0; g < G; g++) {
    const uint32_t c0 = (uint32_t)((uint64_t)k * g / G), c1 = (uint32_t)((uint64_t)k * (g + 1) / G);
    const uint64_t r0 = (uint64_t)c0 * sz, r1 = c1 == k ? n : (uint64_t)c1 * sz;
    ChunkGeo geo;
    geo.k = c1 - c0; geo.sz = sz; geo.rows_last = c1 == k ? rows_last : sz;
    geo.payload_bytes = n ? (uint64_t)((double)data_len * (double)(r1 - r0) / (double)n) : 0;   // (the offsets live on the device)
    rh_opts o = opts;
    o.stream = g % NS == 0 ? (void*)caller : (void*)extra[g % NS - 1];
    o.device = device; o.chunk_rows = 0; o.flags &= ~RH_ASYNC;
    auto res = std::make_unique<rh_device_result>();
    auto call = std::make_unique<DeviceDecode>(s, d_data, d_offsets + r0, data_len, r1 - r0, (uint64_t)geo.k, &o, false, &geo, *res);
    call->async = true;
    if (stagger) {
      call->start_after = prev_sized;
      if (g + 1 < G) {
        prev_sized = pooled_event(device);
        parent->join_events.push_back(prev_sized);
        call->sized = prev_sized;
      }
    }
    try {
      call->enqueue();
      if (call->fused) {
        res->pending = std::move(call);
      } else {
        call->finish();
      }
    } catch (...) {
      call->drain();
      throw;                      // (the groups already enqueued drain in the parent's destructor)
    }
    parent->part_chunk0.push_back(c0);
    parent->parts.push_back(std::move(res));
  }
  // join: the caller's stream continues behind every group
  for (hipStream_t x : extra) {
    hipEvent_t e = pooled_event(device);
    parent->join_events.push_back(e);
    HIPCHK(hipEventRecord(e, x));
    HIPCHK(hipStreamWaitEvent(caller, e, 0));
  }
  if (!async) settle(parent.get());
  return parent.release();
}

// The host's half of an asynchronous call (RH_ASYNC): wait for the stream, check for a malformed record, retry with an
// exact arena if the reserved one was too small, fall back to the generic kernels if a child row domain needs 64-bit
// indexing.  Throws what the synchronous call would have thrown; a failed result stays failed.
void settle(rh_device_result* r) {
  if (r->fail) std::rethrow_exception(r->fail);
  if (!r->parts.empty()) {
    // groups are settled in chunk order: the first failure is the lowest failing group's, i.e. the lowest malformed
    // record of the call (the in-order join of deserialize.rs:115-119)
    try {
      for (auto& p : r->parts) settle(p.get());
    } catch (...) {
      r->fail = std::current_exception();
      throw;
    }
    return;
  }
  if (!r->pending) return;
  std::unique_ptr<DeviceDecode> call = std::move(r->pending);
  // the call again, synchronously, on another form: the two-pass form (a single-pass call that outgrew a capacity) or the
  // generic kernels (a child row domain beyond 32-bit indexing -- which the two-pass repeat may itself run into)
  auto rerun = [&](int add_flags, bool generic) {
    call->drain();
    rh_opts o = call->opts;
    o.flags = generic ? ((o.flags & ~(3 | RH_ASYNC)) | RH_KERNEL_GENERIC) : ((o.flags & ~RH_ASYNC) | add_flags);
    rh_stats st2;
    std::memset(&st2, 0, sizeof st2);
    std::unique_ptr<rh_device_result> r2;
    try {
      r2.reset(decode_device_impl1(call->s, call->d_data, call->d_offsets, call->data_len, call->n, call->num_chunks, &o,
                                   call->want_stats ? &st2 : nullptr, call->geo));
    } catch (const NeedWideIndex&) {
      if (generic) throw;
      count(RH_CTR_WIDE_FALLBACKS);
      o.flags = (o.flags & ~3) | RH_KERNEL_GENERIC;
      r2.reset(decode_device_impl1(call->s, call->d_data, call->d_offsets, call->data_len, call->n, call->num_chunks, &o,
                                   call->want_stats ? &st2 : nullptr, call->geo));
    }
    call.reset();                                  // (its reference to *r ends here)
    r->arena = std::move(r2->arena);
    r->arena_host = r2->arena_host;
    r->arena_bytes = r2->arena_bytes;
    r->buf_off = std::move(r2->buf_off); r->buf_size = std::move(r2->buf_size); r->dom_rows = std::move(r2->dom_rows);
    r->data_bytes = std::move(r2->data_bytes); r->nullcount = std::move(r2->nullcount); r->layout_bytes = std::move(r2->layout_bytes);
    r->output_bytes = r2->output_bytes; r->tables_done = r2->tables_done;
    r->k = r2->k; r->sz = r2->sz; r->rows_last = r2->rows_last;
    if (r2->has_stats || st2.records) { r->st = st2; r->has_stats = true; }
  };
  try {
    try {
      call->finish();
      if (call->want_stats) { r->st = call->st; r->has_stats = true; }
    } catch (const NeedTwoPass&) {
      rerun(RH_INTERNAL_TWO_PASS, false);
    } catch (const NeedWideIndex&) {
      count(RH_CTR_WIDE_FALLBACKS);
      rerun(0, true);
    } catch (const NeedRanged&) {
      count(RH_CTR_RANGED_RETRIES);
      rerun(0, true);
    }
  } catch (...) {
    if (call) call->drain();
    r->arena.release();
    r->fail = std::current_exception();
    throw;
  }
}

rh_device_result* decode_device_impl(rh_schema* s, const uint8_t* d_data, const uint64_t* d_offsets, uint64_t data_len,
                                     uint64_t n, uint64_t num_chunks, const rh_opts* opts, rh_stats* stats,
                                     const ChunkGeo* geo) {
  try {
  try {
    // in-call overlap (decode_device_split): a large call deals its chunk groups to internal streams
    const long G = env_long("RUHVRO_HIP_INTERNAL_STREAMS", kInternalStreamsDefault, 1, 8);
    const long groups_env = env_long("RUHVRO_HIP_SPLIT_GROUPS", 0, 0, 64);      // (> 1 with one stream: the groups run back to back)
    if ((G > 1 || groups_env > 1) && !geo && !stats && n >= (uint64_t)env_long("RUHVRO_HIP_SPLIT_MIN", kSplitMinDefault, 1, 1l << 40) &&
        (!opts || (opts->flags & 3) != RH_KERNEL_GENERIC) && s->arena_ratio.load() > 0 && env_long("RUHVRO_HIP_TWO_SYNC", 0, 0, 1) == 0 &&
        env_long("RUHVRO_HIP_ARENA_PERMILLE", -1, 0, 1000000) < 0) {
      const rh_opts o = opts ? *opts : default_opts();
      uint64_t k64 = rh_clamp_chunks(n, num_chunks), sz = n / std::max<uint64_t>(k64, 1);
      bool ok = true;
      if (o.chunk_rows) {        // explicit geometry: validated by the unsplit path when it does not hold
        ok = num_chunks >= 1 && num_chunks <= 0xFFFFFFFFull && (num_chunks - 1) <= n / o.chunk_rows &&
             !(n > 0 && n == (num_chunks - 1) * o.chunk_rows && num_chunks > 1);
        k64 = num_chunks; sz = o.chunk_rows;
      }
      const unsigned g = (unsigned)std::min<uint64_t>((uint64_t)G, k64);
      if (ok && (g > 1 || (groups_env > 1 && k64 > 1)) && !((uintptr_t)d_data & 15)) {
        const uint64_t rows_last = n - (k64 - 1) * sz;
        try {
          if (rh_device_result* r = decode_device_split(s, d_data, d_offsets, data_len, n, o, (uint32_t)k64, sz, rows_last, g)) return r;
        } catch (const NeedWideIndex&) {
          // (a child row domain beyond 32-bit indexing: the whole call goes to the generic kernels below, unsplit)
          count(RH_CTR_WIDE_FALLBACKS);
          rh_opts o2 = o;
          o2.flags = (o.flags & ~3) | RH_KERNEL_GENERIC;
          return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o2, stats, geo);
        }
      }
    }
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, opts, stats, geo);
  } catch (const NeedTwoPass&) {
    rh_opts o = default_opts();
    if (opts) o = *opts;
    o.flags |= RH_INTERNAL_TWO_PASS;
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o, stats, geo);
  }
  } catch (const NeedWideIndex&) {
    count(RH_CTR_WIDE_FALLBACKS);
    rh_opts o = default_opts();
    if (opts) o = *opts;
    o.flags = RH_KERNEL_GENERIC;
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o, stats, geo);
  } catch (const NeedRanged&) {      // tiles past the LDS window, no ranged pair yet: this call on the generic kernels (the pair compiles in the background)
    count(RH_CTR_RANGED_RETRIES);
    rh_opts o = default_opts();
    if (opts) o = *opts;
    o.flags = (o.flags & ~(3 | RH_ASYNC)) | RH_KERNEL_GENERIC;
    return decode_device_impl1(s, d_data, d_offsets, data_len, n, num_chunks, &o, stats, geo);
  }
}


}  // namespace rhe
