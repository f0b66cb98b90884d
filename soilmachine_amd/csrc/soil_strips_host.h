// soil_strips_host.h -- column strips of ONE map over several devices, driven from C++ inside the library (included by soilmx.hip).
//
// north_star: "Shard the heightmap across up to 8 MI355X as column strips with RCCL halo exchange ... over xGMI". The schedule
// (INTERIOR / SEAM chunks, soil_batch.h BStrips) and the step functions smx_d_* are those of round 2; what moves here is the
// DRIVER (round 2: a Python driver, now tests/strips_ref.py) and the TRANSPORT: halo columns and migrating particles stay in device memory from
// the pack kernel to the unpack kernel on the other rank --
//   RcclTransport   ncclSend / ncclRecv / ncclAllReduce / ncclAllGather on this context's stream (librccl is loaded with dlopen:
//                   the library itself has no link-time dependency on it). One message = a 32-byte header (sizes) followed by up
//                   to three payloads straight out of / into the exchange buffers.
//   HostTransport   caller-supplied callbacks moving host buffers (loop-back threads, torch.distributed gloo, MPI ...): the
//                   payloads are staged through pinned memory. The bit-identity tests run on this one (RCCL refuses two ranks
//                   on one device; tests/test_gpu_strips.py), the protocol above it is the same code.
// Reference anchor of the partition: Layermap index x*dim.y + y (source/layermap.h:151) -- an x-strip is one contiguous block.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct XBundle { uint32_t nrec = 0, nbytes = 0, npart = 0, rs = 0, pad[4] = {0, 0, 0, 0}; };   // the 32-byte header of a seam message
static_assert(sizeof(XBundle) == 32, "seam message header");

struct StripTransport {
  int rank = 0, world = 1;
  uint64_t bytes_sent = 0, messages = 0;
  uint64_t host_syncs = 0, groups = 0;                        // stream synchronisations the transport itself asked for; ncclGroupStart/End pairs (smx_strips_sync_stats)
  virtual ~StripTransport() {}
  // ONE seam exchange of this rank: `out` goes to send_peer, `in` comes from recv_peer (-1: this rank has no such neighbour). Default -- blocking host
  // callbacks --: send, then receive; ranks are visited left to right, so the dependent operations resolve from rank 0 upward and cannot deadlock.
  // The RCCL transport posts both directions together (round 6): no rank's receive waits behind its own send any more.
  virtual int exchange(smx_ctx* ctx, int send_peer, const XBundle* out, int recv_peer, XBundle* in) {
    if (send_peer >= 0 && send_bundle(ctx, send_peer, *out)) return -1;
    if (recv_peer >= 0 && recv_bundle(ctx, recv_peer, *in)) return -1;
    return 0;
  }
  // payload of a send: ctx->d_xoffs (4*nrec bytes), ctx->d_xdata (nbytes), ctx->d_xpart (npart*rs); of a receive: d_roffs, d_rdata, d_rpart
  virtual int send_bundle(smx_ctx* ctx, int peer, const XBundle& b) = 0;
  virtual int recv_bundle(smx_ctx* ctx, int peer, XBundle& b) = 0;
  virtual int allreduce_sum(smx_ctx* ctx, uint64_t& v) = 0;
  virtual int allreduce_min_u32(smx_ctx* ctx, uint32_t* a, size_t n) = 0;                  // host array, in place
  virtual int allgather_children(smx_ctx* ctx, const BChild* mine, uint32_t n, std::vector<BChild>& all) = 0;   // rank order
};

// ------------------------------------------------------------------------------------------------ RCCL
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr; decltype(&ncclAllReduce) AllReduce = nullptr; decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    // one RCCL per process: a host that already carries one (PyTorch bundles its own librccl.so) is asked for that copy first
    for (const char* name : {"librccl.so", "librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }
    if (!lib) for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) { err = "RCCL transport: cannot load librccl.so"; return false; }
#define SMX_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) { err = "RCCL transport: librccl.so lacks nccl" #f; return false; }
    SMX_SYM(GetUniqueId) SMX_SYM(CommInitRank) SMX_SYM(CommDestroy) SMX_SYM(Send) SMX_SYM(Recv) SMX_SYM(AllReduce) SMX_SYM(AllGather) SMX_SYM(GroupStart) SMX_SYM(GroupEnd)
    SMX_SYM(GetErrorString)
#undef SMX_SYM
    return true;
  }
};
static RcclApi g_rccl;
#define NCCLCHK(call)                                                                                \
  do {                                                                                               \
    ncclResult_t r_ = (call);                                                                        \
    if (r_ != ncclSuccess) { ctx->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return -1; } \
  } while (0)

struct RcclTransport : StripTransport {
  ncclComm_t comm = nullptr;
  XBundle* d_hdr = nullptr;                                   // [2] device staging of the headers (send, receive)
  unsigned long long* d_coll = nullptr;                       // small device scratch of the collectives
  void* d_gather = nullptr; size_t gather_cap = 0; void* d_mine = nullptr; size_t mine_cap = 0;
  XBundle* h_hdr = nullptr; uint32_t hdr_slot = 0;            // pinned ring of outgoing headers: the async copy reads its slot when the stream gets there,
  static constexpr uint32_t HDR_RING = 64;                    // long after send_bundle's caller has dropped its own (stack) header; every chunk synchronises
                                                              // the stream at least once (recv_bundle, the live-count all-reduce), so at most a few slots are in flight
  ~RcclTransport() override { if (comm) g_rccl.CommDestroy(comm); hipFree(d_hdr); hipFree(d_coll); hipFree(d_gather); hipFree(d_mine); if (h_hdr) hipHostFree(h_hdr); }
  int send_bundle(smx_ctx* ctx, int peer, const XBundle& b) override {
    hipStream_t st = ctx->stream;
    if (!h_hdr) HIPCHK(hipHostMalloc(&h_hdr, HDR_RING * sizeof(XBundle)));
    XBundle* slot = &h_hdr[hdr_slot++ % HDR_RING];
    *slot = b;
    HIPCHK(hipMemcpyAsync(&d_hdr[0], slot, sizeof(b), hipMemcpyHostToDevice, st));
    // same shape on both sides: the header as an operation of its own (the receiver needs it before it can post the rest),
    // then ONE group with the payloads -- point-to-point operations between two ranks pair up in the order they were issued
    NCCLCHK(g_rccl.Send(&d_hdr[0], sizeof(XBundle), ncclUint8, peer, comm, st));
    NCCLCHK(g_rccl.GroupStart());
    if (b.nrec) { NCCLCHK(g_rccl.Send(ctx->d_xoffs, 4ull * b.nrec, ncclUint8, peer, comm, st)); NCCLCHK(g_rccl.Send(ctx->d_xdata, b.nbytes, ncclUint8, peer, comm, st)); }
    if (b.npart) NCCLCHK(g_rccl.Send(ctx->d_xpart, (size_t)b.npart * b.rs, ncclUint8, peer, comm, st));
    NCCLCHK(g_rccl.GroupEnd());
    bytes_sent += sizeof(XBundle) + 4ull * b.nrec + b.nbytes + (uint64_t)b.npart * b.rs; messages++;
    return 0;
  }
  int recv_bundle(smx_ctx* ctx, int peer, XBundle& b) override {
    hipStream_t st = ctx->stream;
    NCCLCHK(g_rccl.Recv(&d_hdr[1], sizeof(XBundle), ncclUint8, peer, comm, st));
    HIPCHK(hipMemcpyAsync(&b, &d_hdr[1], sizeof(b), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); host_syncs++;           // the payload sizes decide the receives that follow
    if (b.nrec > ctx->xoffs_cap || b.nbytes > ctx->xdata_cap || (uint64_t)b.npart * b.rs > ctx->xpart_cap) { ctx->err = "strips: incoming seam message exceeds the exchange buffers"; return -2; }
    NCCLCHK(g_rccl.GroupStart());
    if (b.nrec) { NCCLCHK(g_rccl.Recv(ctx->d_roffs, 4ull * b.nrec, ncclUint8, peer, comm, st)); NCCLCHK(g_rccl.Recv(ctx->d_rdata, b.nbytes, ncclUint8, peer, comm, st)); }
    if (b.npart) NCCLCHK(g_rccl.Recv(ctx->d_rpart, (size_t)b.npart * b.rs, ncclUint8, peer, comm, st));
    NCCLCHK(g_rccl.GroupEnd());
    return 0;
  }
  // Both directions of a seam exchange in TWO grouped operations (round 6): {header out, header in}, one stream synchronisation
  // for the incoming sizes, {payloads out, payloads in}. Rounds 3-5 issued send (header, payloads) and then receive (header, sync, payloads) one after
  // the other on the stream: rank r's receive from r + 1 sat behind its own send to r - 1, whose completion needs r - 1's receive, which sat behind
  // r - 1's send ... -- an (N - 1)-hop chain of dependent point-to-point kernels per exchange. Inside one ncclGroup the operations progress together.
  int exchange(smx_ctx* ctx, int send_peer, const XBundle* out, int recv_peer, XBundle* in) override {
    hipStream_t st = ctx->stream;
    if (send_peer < 0 && recv_peer < 0) return 0;
    if (!h_hdr) HIPCHK(hipHostMalloc(&h_hdr, HDR_RING * sizeof(XBundle)));
    XBundle* hin = &h_hdr[hdr_slot++ % HDR_RING];
    if (send_peer >= 0) {
      XBundle* slot = &h_hdr[hdr_slot++ % HDR_RING];
      *slot = *out;
      HIPCHK(hipMemcpyAsync(&d_hdr[0], slot, sizeof(XBundle), hipMemcpyHostToDevice, st));
    }
    NCCLCHK(g_rccl.GroupStart());
    if (send_peer >= 0) NCCLCHK(g_rccl.Send(&d_hdr[0], sizeof(XBundle), ncclUint8, send_peer, comm, st));
    if (recv_peer >= 0) NCCLCHK(g_rccl.Recv(&d_hdr[1], sizeof(XBundle), ncclUint8, recv_peer, comm, st));
    NCCLCHK(g_rccl.GroupEnd());
    groups++;
    XBundle bi;
    if (recv_peer >= 0) {
      HIPCHK(hipMemcpyAsync(hin, &d_hdr[1], sizeof(XBundle), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st)); host_syncs++;        // the payload sizes decide the receives that follow
      bi = *hin; *in = bi;
      if (bi.nrec > ctx->xoffs_cap || bi.nbytes > ctx->xdata_cap || (uint64_t)bi.npart * bi.rs > ctx->xpart_cap) { ctx->err = "strips: incoming seam message exceeds the exchange buffers"; return -2; }
    }
    const bool so = send_peer >= 0 && (out->nrec || out->npart), si = recv_peer >= 0 && (bi.nrec || bi.npart);
    if (so || si) {
      NCCLCHK(g_rccl.GroupStart());
      if (so) {
        if (out->nrec) { NCCLCHK(g_rccl.Send(ctx->d_xoffs, 4ull * out->nrec, ncclUint8, send_peer, comm, st)); NCCLCHK(g_rccl.Send(ctx->d_xdata, out->nbytes, ncclUint8, send_peer, comm, st)); }
        if (out->npart) NCCLCHK(g_rccl.Send(ctx->d_xpart, (size_t)out->npart * out->rs, ncclUint8, send_peer, comm, st));
      }
      if (si) {
        if (bi.nrec) { NCCLCHK(g_rccl.Recv(ctx->d_roffs, 4ull * bi.nrec, ncclUint8, recv_peer, comm, st)); NCCLCHK(g_rccl.Recv(ctx->d_rdata, bi.nbytes, ncclUint8, recv_peer, comm, st)); }
        if (bi.npart) NCCLCHK(g_rccl.Recv(ctx->d_rpart, (size_t)bi.npart * bi.rs, ncclUint8, recv_peer, comm, st));
      }
      NCCLCHK(g_rccl.GroupEnd());
      groups++;
    }
    if (send_peer >= 0) { bytes_sent += sizeof(XBundle) + 4ull * out->nrec + out->nbytes + (uint64_t)out->npart * out->rs; messages++; }
    return 0;
  }
  int allreduce_sum(smx_ctx* ctx, uint64_t& v) override {
    hipStream_t st = ctx->stream;
    unsigned long long h = v;
    HIPCHK(hipMemcpyAsync(d_coll, &h, 8, hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllReduce(d_coll, d_coll + 1, 1, ncclUint64, ncclSum, comm, st));
    HIPCHK(hipMemcpyAsync(&h, d_coll + 1, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); host_syncs++;
    v = h;
    return 0;
  }
  int ensure(smx_ctx* ctx, void*& p, size_t& cap, size_t need) {
    if (cap >= need) return 0;
    hipFree(p); p = nullptr; cap = 0;
    HIPCHK(hipMalloc(&p, need)); cap = need;
    return 0;
  }
  int allreduce_min_u32(smx_ctx* ctx, uint32_t* a, size_t n) override {
    if (n == 0) return 0;
    hipStream_t st = ctx->stream;
    if (ensure(ctx, d_mine, mine_cap, n * 4)) return -1;
    HIPCHK(hipMemcpyAsync(d_mine, a, n * 4, hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllReduce(d_mine, d_mine, n, ncclUint32, ncclMin, comm, st));
    HIPCHK(hipMemcpyAsync(a, d_mine, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  int allgather_children(smx_ctx* ctx, const BChild* mine, uint32_t n, std::vector<BChild>& all) override {
    hipStream_t st = ctx->stream;
    std::vector<unsigned long long> cnt((size_t)world, 0ull);
    unsigned long long h = n;
    HIPCHK(hipMemcpyAsync(d_coll, &h, 8, hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllGather(d_coll, d_coll + 8, 1, ncclUint64, comm, st));
    HIPCHK(hipMemcpyAsync(cnt.data(), d_coll + 8, 8ull * world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unsigned long long mx = 0, tot = 0;
    for (unsigned long long c : cnt) { mx = std::max(mx, c); tot += c; }
    all.clear();
    if (tot == 0) return 0;
    const size_t each = (size_t)mx * sizeof(BChild);          // every rank contributes a block padded to the longest list
    if (ensure(ctx, d_mine, mine_cap, each) || ensure(ctx, d_gather, gather_cap, each * world)) return -1;
    if (n) HIPCHK(hipMemcpyAsync(d_mine, mine, (size_t)n * sizeof(BChild), hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllGather(d_mine, d_gather, each, ncclUint8, comm, st));
    std::vector<BChild> buf((size_t)mx * world);
    HIPCHK(hipMemcpyAsync(buf.data(), d_gather, each * world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    all.reserve(tot);
    for (int r = 0; r < world; r++) all.insert(all.end(), buf.begin() + (size_t)r * mx, buf.begin() + (size_t)r * mx + cnt[r]);
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------ caller-supplied host transport
struct HostTransport : StripTransport {
  smx_transport t{};
  uint8_t* h_buf = nullptr; uint64_t h_cap = 0;               // pinned staging of one message
  ~HostTransport() override { if (h_buf) hipHostFree(h_buf); }
  int stage(smx_ctx* ctx, uint64_t need) {
    if (h_cap >= need) return 0;
    if (h_buf) hipHostFree(h_buf);
    h_buf = nullptr; h_cap = 0;
    HIPCHK(hipHostMalloc(&h_buf, need)); h_cap = need;
    return 0;
  }
  int send_bundle(smx_ctx* ctx, int peer, const XBundle& b) override {
    hipStream_t st = ctx->stream;
    const uint64_t total = sizeof(XBundle) + 4ull * b.nrec + b.nbytes + (uint64_t)b.npart * b.rs;
    if (stage(ctx, total)) return -1;
    memcpy(h_buf, &b, sizeof(b));
    uint64_t o = sizeof(XBundle);
    if (b.nrec) { HIPCHK(hipMemcpyAsync(h_buf + o, ctx->d_xoffs, 4ull * b.nrec, hipMemcpyDeviceToHost, st)); o += 4ull * b.nrec;
                  HIPCHK(hipMemcpyAsync(h_buf + o, ctx->d_xdata, b.nbytes, hipMemcpyDeviceToHost, st)); o += b.nbytes; }
    if (b.npart) HIPCHK(hipMemcpyAsync(h_buf + o, ctx->d_xpart, (size_t)b.npart * b.rs, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (t.send(t.user, peer, h_buf, total) != 0) { ctx->err = "strips: transport send failed"; return -1; }
    bytes_sent += total; messages++;
    return 0;
  }
  int recv_bundle(smx_ctx* ctx, int peer, XBundle& b) override {
    hipStream_t st = ctx->stream;
    const uint64_t cap = sizeof(XBundle) + 4ull * ctx->xoffs_cap + ctx->xdata_cap + ctx->xpart_cap;
    if (stage(ctx, std::min<uint64_t>(cap, std::max<uint64_t>(h_cap, 64ull << 20)))) return -1;
    uint64_t got = 0;
    int rc = t.recv(t.user, peer, h_buf, h_cap, &got);
    if (rc == 1) {                                            // (message larger than the staging buffer: the transport kept it; grow and retry)
      if (stage(ctx, got)) return -1;
      rc = t.recv(t.user, peer, h_buf, h_cap, &got);
    }
    if (rc != 0 || got < sizeof(XBundle)) { ctx->err = "strips: transport recv failed"; return -1; }
    memcpy(&b, h_buf, sizeof(b));
    const uint64_t total = sizeof(XBundle) + 4ull * b.nrec + b.nbytes + (uint64_t)b.npart * b.rs;
    if (total != got || b.nrec > ctx->xoffs_cap || b.nbytes > ctx->xdata_cap || (uint64_t)b.npart * b.rs > ctx->xpart_cap) { ctx->err = "strips: malformed seam message"; return -2; }
    uint64_t o = sizeof(XBundle);
    if (b.nrec) { HIPCHK(hipMemcpyAsync(ctx->d_roffs, h_buf + o, 4ull * b.nrec, hipMemcpyHostToDevice, st)); o += 4ull * b.nrec;
                  HIPCHK(hipMemcpyAsync(ctx->d_rdata, h_buf + o, b.nbytes, hipMemcpyHostToDevice, st)); o += b.nbytes; }
    if (b.npart) HIPCHK(hipMemcpyAsync(ctx->d_rpart, h_buf + o, (size_t)b.npart * b.rs, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                         // (the staging buffer is reused by the next message)
    return 0;
  }
  int allreduce_sum(smx_ctx* ctx, uint64_t& v) override { if (t.allreduce_sum_u64(t.user, &v) != 0) { ctx->err = "strips: transport allreduce failed"; return -1; } return 0; }
  int allreduce_min_u32(smx_ctx* ctx, uint32_t* a, size_t n) override { if (n && t.allreduce_min_u32(t.user, a, n) != 0) { ctx->err = "strips: transport allreduce failed"; return -1; } return 0; }
  int allgather_children(smx_ctx* ctx, const BChild* mine, uint32_t n, std::vector<BChild>& all) override {
    std::vector<uint64_t> sizes((size_t)world, 0);
    uint64_t mx = (uint64_t)n * sizeof(BChild);
    if (t.allreduce_max_u64(t.user, &mx) != 0) { ctx->err = "strips: transport allreduce failed"; return -1; }
    all.clear();
    if (mx == 0) return 0;
    std::vector<uint8_t> buf((size_t)mx * world);
    if (t.allgather(t.user, mine, (uint64_t)n * sizeof(BChild), buf.data(), mx, sizes.data()) != 0) { ctx->err = "strips: transport allgather failed"; return -1; }
    for (int r = 0; r < world; r++) {
      const BChild* p = reinterpret_cast<const BChild*>(buf.data() + (size_t)r * mx);
      all.insert(all.end(), p, p + sizes[r] / sizeof(BChild));
    }
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------ the driver (one rank)
struct StripState {
  StripTransport* T = nullptr;
  BStrips geo{};
  int x0 = 0, x1 = 0;
  uint64_t chunks = 0, seam_chunks = 0, generations = 0;
};

static int xbuf_alloc_strips(smx_ctx* ctx) {                  // exchange buffers of both directions + the particle regions
  if (xbuf_alloc(ctx)) return -1;
  if (!ctx->d_rdata) {
    HIPCHK(hipMalloc(&ctx->d_rdata, ctx->xdata_cap)); HIPCHK(hipMalloc(&ctx->d_roffs, (size_t)ctx->xoffs_cap * 4));
  }
  const uint64_t need = (uint64_t)std::max<uint32_t>(ctx->batch_cap, 4096u) * (4 + std::max(sizeof(BWater), sizeof(BWind)));
  if (ctx->xpart_cap < need) {
    hipFree(ctx->d_xpart); hipFree(ctx->d_rpart); ctx->d_xpart = ctx->d_rpart = nullptr; ctx->xpart_cap = 0;
    HIPCHK(hipMalloc(&ctx->d_xpart, need)); HIPCHK(hipMalloc(&ctx->d_rpart, need)); ctx->xpart_cap = need;
  }
  return 0;
}
// dirty halo columns of [cx0, cx1) and (optionally) the running particles of [px0, px1): packed on the device; the header says how much
// (both pack kernels run back to back, their two cursors come home with ONE stream synchronisation -- rounds 3-5: one per kernel)
static int st_pack_band(smx_ctx* ctx, StripState& S, int cx0, int cx1, int px0, int px1, bool particles, XBundle& b) {
  hipStream_t st = ctx->stream;
  if (batch_alloc(ctx, 1, false) || xbuf_alloc_strips(ctx)) return -1;
  b = XBundle();
  if (cx0 < 0) cx0 = 0; if (cx1 > ctx->cfg.dimx) cx1 = ctx->cfg.dimx;
  const int gx0 = cx0 >> 2, gx1 = (cx1 + 3) >> 2, gty = (ctx->cfg.dimy + 3) >> 2;
  const long long nt = (long long)(gx1 - gx0) * gty;
  XCursor c[2] = {};
  HIPCHK(hipMemsetAsync(ctx->d_xcur, 0, 2 * sizeof(XCursor), st));
  if (nt > 0) hipLaunchKernelGGL(k_strip_pack_columns, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, st, ctx->d, ctx->bs, gx0, gx1, ctx->d_xdata, ctx->xdata_cap, ctx->d_xoffs, ctx->xoffs_cap, ctx->d_xcur);
  const bool parts = particles && ctx->d_nslots;
  if (parts) {
    const uint32_t nslots = ctx->d_nslots;
    if (ctx->d_wind) hipLaunchKernelGGL(k_strip_pack_particles<true>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, ctx->bs, nslots, px0, px1, ctx->d_xpart, ctx->xpart_cap, ctx->d_xcur + 1);
    else hipLaunchKernelGGL(k_strip_pack_particles<false>, dim3((nslots + 255u) / 256u), dim3(256), 0, st, ctx->bs, nslots, px0, px1, ctx->d_xpart, ctx->xpart_cap, ctx->d_xcur + 1);
  }
  HIPCHK(hipMemcpyAsync(c, ctx->d_xcur, 2 * sizeof(XCursor), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st)); S.T->host_syncs++;
  if (c[0].lost) { ctx->err = "strips: exchange buffer too small for the halo columns"; return -2; }
  if (c[1].lost) { ctx->err = "strips: exchange buffer too small for the migrating particles"; return -2; }
  b.nrec = c[0].nrec; b.nbytes = (uint32_t)c[0].bytes;
  b.rs = (uint32_t)(4 + (ctx->d_wind ? sizeof(BWind) : sizeof(BWater)));
  if (parts) b.npart = c[1].nrec;
  (void)S;
  return 0;
}
static int st_unpack_band(smx_ctx* ctx, const XBundle& b) {
  hipStream_t st = ctx->stream;
  if (b.nrec) {
    hipLaunchKernelGGL(k_strip_unpack_columns, dim3((b.nrec + 63u) / 64u), dim3(64), 0, st, ctx->d, ctx->bs, ctx->d_rdata, ctx->d_roffs, b.nrec, b.nbytes);
    hipLaunchKernelGGL(k_batch_merge_freed, dim3(1), dim3(256), 0, st, ctx->d, ctx->bs);
  }
  if (b.npart) {
    if (b.rs != 4 + (ctx->d_wind ? sizeof(BWind) : sizeof(BWater))) { ctx->err = "strips: particle records of the wrong phase"; return -2; }
    if (ctx->d_wind) hipLaunchKernelGGL(k_strip_unpack_particles<true>, dim3((b.npart + 255u) / 256u), dim3(256), 0, st, ctx->bs, ctx->d_rpart, b.npart);
    else hipLaunchKernelGGL(k_strip_unpack_particles<false>, dim3((b.npart + 255u) / 256u), dim3(256), 0, st, ctx->bs, ctx->d_rpart, b.npart);
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// one seam exchange: what this rank packs goes to `send_peer`, what `recv_peer` packed comes in and is unpacked (StripTransport::exchange)
static int st_exchange(smx_ctx* ctx, StripState& S, int send_peer, int cx0, int cx1, int px0, int px1, bool particles, int recv_peer) {
  if (batch_alloc(ctx, 1, false) || xbuf_alloc_strips(ctx)) return -1;
  XBundle out, in;
  if (send_peer >= 0 && st_pack_band(ctx, S, cx0, cx1, px0, px1, particles, out)) return -1;
  if (S.T->exchange(ctx, send_peer, &out, recv_peer, &in)) return -1;
  if (recv_peer >= 0 && st_unpack_band(ctx, in)) return -1;
  return 0;
}
// A rank holds, besides its strip, the b columns right of its right boundary (the seam zone it runs).
static int st_band_to_left(smx_ctx* ctx, StripState& S, bool particles) {     // band [x0, x0+b) of this strip -> the rank on the left (which runs that seam)
  const int r = S.T->rank, n = S.T->world;
  return st_exchange(ctx, S, r > 0 ? r - 1 : -1, S.x0, S.x0 + S.geo.b, S.x0, S.x0 + S.geo.b, particles, r < n - 1 ? r + 1 : -1);
}
static int st_band_to_right(smx_ctx* ctx, StripState& S, bool particles) {    // the halo [x1, x1+b) goes home, with every particle that sits beyond x1
  const int r = S.T->rank, n = S.T->world;
  return st_exchange(ctx, S, r < n - 1 ? r + 1 : -1, S.x1, S.x1 + S.geo.b, S.x1, ctx->cfg.dimx, particles, r > 0 ? r - 1 : -1);
}
static int st_gather_sorted(smx_ctx* ctx, StripState& S, std::vector<BChild>& kids) {   // this rank's new children -> everybody's, in key order
  uint32_t nc = 0;
  std::vector<BChild> mine;
  if (d_gen_end_impl(ctx, nullptr, 0, &nc, &mine)) return -1;
  if (S.T->allgather_children(ctx, mine.data(), nc, kids)) return -1;
  std::stable_sort(kids.begin(), kids.end(), [](const BChild& a, const BChild& b) { return a.key < b.key; });
  return 0;
}
static int st_phase(smx_ctx* ctx, StripState& S, bool wind, uint32_t n, std::vector<BChild> kids, bool have_kids) {
  uint32_t nslots = have_kids ? (uint32_t)kids.size() : n;
  std::vector<uint32_t> contains;
  while (nslots > 0) {
    contains.assign(have_kids ? nslots : 0, 0u);
    if (smx_d_gen_begin(ctx, wind ? 1 : 0, nslots, have_kids ? kids.data() : nullptr, have_kids ? contains.data() : nullptr)) return -1;
    // (rounds 2-4: a min-all-reduce of the children's `contains`, read by the owner of each child's random cell; the records carry it since round 5)
    uint32_t local = 0;
    if (smx_d_gen_spawn(ctx, have_kids ? contains.data() : nullptr, &local)) return -1;
    uint64_t nlive = local;
    if (S.T->allreduce_sum(ctx, nlive)) return -1;
    uint64_t stalled = 0, last = nlive + 1;                   // chunks in a row that ended no particle anywhere
    while (nlive > 0) {
      const bool seam = S.geo.n > 1 && smx_d_next_phase(ctx) == STRIP_SEAM;
      if (seam && st_band_to_left(ctx, S, true)) return -1;
      if (smx_d_chunk(ctx, (uint32_t)nlive, &local)) return -1;
      if (seam) { if (st_band_to_right(ctx, S, true)) return -1; S.seam_chunks++; }
      S.chunks++;
      nlive = local;
      if (S.T->allreduce_sum(ctx, nlive)) return -1;
      stalled = nlive == last ? stalled + 1 : 0; last = nlive;
      if (stalled > (1u << 14)) {                              // (the longest particle lives ~13 600 steps = ~425 chunks of progress for somebody)
        ctx->err = "strips: " + std::to_string(nlive) + " particles made no progress for 16 384 chunks (" + std::to_string(ctx->h_bctrl[BC_RANGE_REJECT]) +
                   " reservations of this rank reached past the columns it holds: strip geometry and schedule margins do not fit)";
        return -8;
      }
    }
    S.generations++;
    if (wind) { uint32_t nc = 0; std::vector<BChild> none; if (d_gen_end_impl(ctx, nullptr, 0, &nc, &none)) return -1; break; }
    if (st_gather_sorted(ctx, S, kids)) return -1;
    have_kids = true;
    nslots = (uint32_t)kids.size();
  }
  if (S.geo.n > 1 && st_band_to_left(ctx, S, false)) return -1;   // the halo is current again (frequency planes, next phase)
  return 0;
}
static int st_grid(smx_ctx* ctx, StripState& S) {
  const int r = S.T->rank, n = S.geo.n;
  if (smx_d_grid_begin(ctx)) return -1;
  if (smx_d_grid_sweep_cols(ctx, STRIP_INTERIOR, S.x0, S.x1)) return -1;
  if (n > 1) {
    if (st_band_to_left(ctx, S, false)) return -1;
    if (r < n - 1 && smx_d_grid_sweep_cols(ctx, STRIP_SEAM, S.x1 - S.geo.b, S.x1 + S.geo.b)) return -1;
    if (st_band_to_right(ctx, S, false)) return -1;
  }
  std::vector<BChild> kids;
  if (st_gather_sorted(ctx, S, kids)) return -1;
  if (!kids.empty()) return st_phase(ctx, S, false, 0, std::move(kids), true);
  if (n > 1 && st_band_to_left(ctx, S, false)) return -1;
  return 0;
}

static void strips_free(smx_ctx* ctx) { if (ctx->strip) { delete ctx->strip->T; delete ctx->strip; ctx->strip = nullptr; } }
extern "C" {
static int strips_attach_common(smx_ctx* ctx, StripTransport* T, int rank, int world, int inset, int seam_halfwidth) {
  if (smx_set_batch_strips(ctx, world, inset, seam_halfwidth)) { delete T; return -2; }
  strips_free(ctx); ctx->strip = new StripState();
  T->rank = rank; T->world = world;
  ctx->strip->T = T;
  ctx->strip->geo = batch_make_strips(ctx->cfg.dimx, world, inset, seam_halfwidth);
  ctx->strip->x0 = ctx->strip->geo.x[rank]; ctx->strip->x1 = ctx->strip->geo.x[rank + 1];
  {   // a strip context (smx_create_strip) must hold its strip, the seam zone right of it and a few columns of slack on either side
    const int need_lo = rank > 0 ? ctx->strip->x0 - 4 : 0, need_hi = rank < world - 1 ? std::min(ctx->cfg.dimx, ctx->strip->x1 + seam_halfwidth + 4) : ctx->cfg.dimx;
    if (ctx->d.x_lo > need_lo || ctx->d.x_hi < need_hi) { ctx->err = "smx_strips_attach: the context does not hold its strip + halo (smx_create_strip range too small)"; strips_free(ctx); return -2; }
  }
  return smx_d_set_own(ctx, ctx->strip->x0, ctx->strip->x1);
}
int smx_strips_rccl_unique_id(void* out128) {
  std::string err;
  if (!out128 || !g_rccl.load(err)) return -1;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return -1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
  memcpy(out128, &id, 128);
  return 0;
}
int smx_strips_attach_rccl(smx_ctx* ctx, const void* unique_id128, int32_t rank, int32_t world, int32_t inset, int32_t seam_halfwidth) {
  if (!unique_id128 || rank < 0 || rank >= world || world < 1 || world > 8) { ctx->err = "smx_strips_attach_rccl: 1..8 ranks"; return -2; }
  if (!g_rccl.load(ctx->err)) return -3;
  HIPCHK(hipSetDevice(ctx->cfg.device));
  RcclTransport* T = new RcclTransport();
  ncclUniqueId id; memcpy(&id, unique_id128, 128);
  { ncclResult_t r_ = g_rccl.CommInitRank(&T->comm, world, id, rank); if (r_ != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r_); delete T; return -1; } }
  if (hipMalloc(&T->d_hdr, 2 * sizeof(XBundle)) != hipSuccess || hipMalloc(&T->d_coll, 64 * 8) != hipSuccess) { ctx->err = "smx_strips_attach_rccl: out of device memory"; delete T; return -1; }
  return strips_attach_common(ctx, T, rank, world, inset, seam_halfwidth);
}
int smx_strips_attach(smx_ctx* ctx, const smx_transport* t, int32_t rank, int32_t world, int32_t inset, int32_t seam_halfwidth) {
  if (!t || !t->send || !t->recv || !t->allreduce_sum_u64 || !t->allreduce_max_u64 || !t->allreduce_min_u32 || !t->allgather || rank < 0 || rank >= world || world < 1 || world > 8) {
    ctx->err = "smx_strips_attach: incomplete transport or rank out of range"; return -2;
  }
  HostTransport* T = new HostTransport();
  T->t = *t;
  return strips_attach_common(ctx, T, rank, world, inset, seam_halfwidth);
}
int smx_strips_detach(smx_ctx* ctx) {
  strips_free(ctx);
  ctx->own_x0 = 0; ctx->own_x1 = -1;
  return smx_set_batch_strips(ctx, 1, 16, 48);
}
// the tick of SoilMachine.cpp:283-329 on this rank's strip; every rank of the transport calls it with the same arguments
int smx_strips_tick(smx_ctx* ctx, int32_t nwater, int32_t nwind, int32_t dowater, int32_t dowind) {
  if (!ctx->strip) { ctx->err = "smx_strips_tick: no transport attached (smx_strips_attach / smx_strips_attach_rccl)"; return -2; }
  if (!is_batched(ctx)) { ctx->err = "smx_strips_tick: column strips run the batched engine (the exact engines are sequential: replicas only)"; return -2; }
  StripState& S = *ctx->strip;
  if (dowater) {
    roctx_range rw("soilmx:strips:water");
    uint32_t left = nwater > 0 ? (uint32_t)nwater : 0u;       // as smx_tick_water: k consecutive top-level generations of n/k
    const uint32_t k = ctx->water_generations < 1 ? 1u : (uint32_t)ctx->water_generations;
    if (ctx->cfg.engine == SMX_ENGINE_RELAXED && ctx->water_stagger > 0 && k > 1u && left > 0u) {   // staggered generations: one phase (smx_tick_water)
      ctx->stagger_phase_n = left;
      const int rc = st_phase(ctx, S, false, left, {}, false);
      ctx->stagger_phase_n = 0u;
      if (rc) return -1;
      left = 0u;
    }
    for (uint32_t q = 0; q < k && left > 0; q++) {
      const uint32_t nq = left / (k - q);
      left -= nq;
      if (nq && st_phase(ctx, S, false, nq, {}, false)) return -1;
    }
    roctx_range rg("soilmx:strips:grid");
    if (st_grid(ctx, S)) return -1;
  }
  if (dowind) { roctx_range rd("soilmx:strips:wind"); if (st_phase(ctx, S, true, nwind > 0 ? (uint32_t)nwind : 0u, {}, false)) return -1; }
  if (dowater) { if (smx_map_frequency(ctx)) return -1; if (smx_reset_frequency(ctx)) return -1; }
  return 0;
}
int smx_strips_sync_stats(smx_ctx* ctx, uint64_t* host_syncs, uint64_t* groups) {   // what the transport itself cost: stream synchronisations, grouped operations
  if (!ctx->strip) return -2;
  if (host_syncs) *host_syncs = ctx->strip->T->host_syncs; if (groups) *groups = ctx->strip->T->groups;
  return 0;
}
// ONE synthetic seam message from this rank to this very rank through the attached transport's exchange() -- header group, size synchronisation, payload group --
// compared byte for byte on the way back. With RCCL a send and a receive of the same rank pair up inside one group, so a single device exercises the
// grouped ncclSend / ncclRecv calls, their stream ordering and the header ring of RcclTransport::exchange that a run on several GPUs depends on (round 6).
int smx_strips_selfcheck(smx_ctx* ctx, uint32_t nrec, uint32_t nbytes, uint32_t npart) {
  if (!ctx->strip) { ctx->err = "smx_strips_selfcheck: no transport attached"; return -2; }
  if (batch_alloc(ctx, 1, false) || xbuf_alloc_strips(ctx)) return -1;
  const uint32_t rs = 4 + (uint32_t)sizeof(BWater);
  if (nrec > ctx->xoffs_cap || nbytes > ctx->xdata_cap || (uint64_t)npart * rs > ctx->xpart_cap) { ctx->err = "smx_strips_selfcheck: larger than the exchange buffers"; return -2; }
  hipStream_t st = ctx->stream;
  StripTransport* T = ctx->strip->T;
  std::vector<uint32_t> offs(nrec); std::vector<uint8_t> data(nbytes), part((size_t)npart * rs);
  uint32_t x = 0x9e3779b9u ^ (nrec * 31u + nbytes * 7u + npart);
  auto nxt = [&x]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
  for (auto& v : offs) v = nxt(); for (auto& v : data) v = (uint8_t)nxt(); for (auto& v : part) v = (uint8_t)nxt();
  if (nrec) HIPCHK(hipMemcpyAsync(ctx->d_xoffs, offs.data(), 4ull * nrec, hipMemcpyHostToDevice, st));
  if (nbytes) HIPCHK(hipMemcpyAsync(ctx->d_xdata, data.data(), nbytes, hipMemcpyHostToDevice, st));
  if (npart) HIPCHK(hipMemcpyAsync(ctx->d_xpart, part.data(), part.size(), hipMemcpyHostToDevice, st));
  if (nrec) HIPCHK(hipMemsetAsync(ctx->d_roffs, 0, 4ull * nrec, st)); if (nbytes) HIPCHK(hipMemsetAsync(ctx->d_rdata, 0, nbytes, st)); if (npart) HIPCHK(hipMemsetAsync(ctx->d_rpart, 0, part.size(), st));
  XBundle out, in; out.nrec = nrec; out.nbytes = nbytes; out.npart = npart; out.rs = rs; out.pad[0] = 0x5e1fc4ecu;
  if (T->exchange(ctx, T->rank, &out, T->rank, &in)) return -1;
  HIPCHK(hipStreamSynchronize(st));
  if (memcmp(&in, &out, sizeof(XBundle)) != 0) { ctx->err = "smx_strips_selfcheck: the header came back changed"; return -4; }
  std::vector<uint32_t> roffs(nrec); std::vector<uint8_t> rdata(nbytes), rpart(part.size());
  if (nrec) HIPCHK(hipMemcpy(roffs.data(), ctx->d_roffs, 4ull * nrec, hipMemcpyDeviceToHost));
  if (nbytes) HIPCHK(hipMemcpy(rdata.data(), ctx->d_rdata, nbytes, hipMemcpyDeviceToHost));
  if (npart) HIPCHK(hipMemcpy(rpart.data(), ctx->d_rpart, part.size(), hipMemcpyDeviceToHost));
  if (roffs != offs || rdata != data || rpart != part) { ctx->err = "smx_strips_selfcheck: a payload came back changed"; return -4; }
  return 0;
}
int smx_strips_stats(smx_ctx* ctx, uint64_t* chunks, uint64_t* seam_chunks, uint64_t* generations, uint64_t* bytes_sent, uint64_t* messages) {
  if (!ctx->strip) return -2;
  if (chunks) *chunks = ctx->strip->chunks; if (seam_chunks) *seam_chunks = ctx->strip->seam_chunks; if (generations) *generations = ctx->strip->generations;
  if (bytes_sent) *bytes_sent = ctx->strip->T->bytes_sent; if (messages) *messages = ctx->strip->T->messages;
  return 0;
}
}  // extern "C"
