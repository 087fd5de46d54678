// api_multi_gpu.hip -- C ABI of the z-slab multi-GPU path: attaching a context to a rank and its communicator.
// The exchanges themselves are issued by the engine (comm.hip, loop.hip, engine.hip); the reference has no counterpart (single process).
#include "engine_internal.h"

#include <sys/socket.h>
#include <sys/time.h>
#include <cerrno>
#include <atomic>
#include <algorithm>
#include <thread>

using namespace psge;

// ---- a node-local transport over connected stream sockets (psgsdf_comm_init_sockets): every primitive drains the stream, stages through the host and
// blocks.  For a host whose ranks share one device (RCCL refuses that: the one-GPU rehearsals of `voxelPS --gpus N`) or that has no working RCCL;
// the per-iteration exchanges of the optimisation do not go through it (they are in-kernel) -- set-up, fusion, re-cuts and dumps do.
namespace {
struct SockComm { int rank = 0, n = 1; std::vector<int> fd; };
bool wr_all(int fd, const void* p, size_t n) {
    const char* q = (const char*)p;
    while (n) { const ssize_t k = send(fd, q, n, MSG_NOSIGNAL); if (k < 0) { if (errno == EINTR) continue; return false; } q += k; n -= (size_t)k; }
    return true;
}
bool rd_all(int fd, void* p, size_t n) {
    char* q = (char*)p;
    while (n) { const ssize_t k = recv(fd, q, n, 0); if (k < 0) { if (errno == EINTR) continue; return false; } if (k == 0) return false; q += k; n -= (size_t)k; }
    return true;
}
// in-place sum: rank 0 adds the ranks' vectors in rank order and hands the result back -- every rank gets the same bits
int sock_allreduce(void* user, double* dev, int n, void* stream) {
    SockComm* s = (SockComm*)user;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    std::vector<double> h((size_t)n), t((size_t)n);
    if (hipMemcpy(h.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (s->rank == 0) {
        for (int r = 1; r < s->n; ++r) { if (!rd_all(s->fd[r], t.data(), sizeof(double) * n)) return 1; for (int i = 0; i < n; ++i) h[i] += t[i]; }
        for (int r = 1; r < s->n; ++r) if (!wr_all(s->fd[r], h.data(), sizeof(double) * n)) return 1;
    } else {
        if (!wr_all(s->fd[0], h.data(), sizeof(double) * n) || !rd_all(s->fd[0], h.data(), sizeof(double) * n)) return 1;
    }
    return hipMemcpy(dev, h.data(), sizeof(double) * n, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
// The sends leave on threads of their own -- ONE PER PEER -- while this one receives in list order.  (One sender thread for all peers, round 5, could be
// blocked on a full socket buffer towards a peer that was itself reading somebody else first: with three or more ranks and payloads beyond the socket
// buffers -- psgsdf_rebalance_slabs allows any peer -- such waits can close a cycle, ended only by the send timeout.  With a thread per peer a send to P
// waits for P alone, and P's reads are fed by senders that wait for nobody else: no cycle.  ADVICE r05.)
int sock_sendrecv(void* user, const psgsdf_comm_xfer* sends, int ns, const psgsdf_comm_xfer* recvs, int nr, void* stream) {
    SockComm* s = (SockComm*)user;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    std::vector<std::vector<char>> out((size_t)ns), in((size_t)nr);
    for (int i = 0; i < ns; ++i) {
        if (sends[i].peer < 0 || sends[i].peer >= s->n || sends[i].peer == s->rank) return 1;
        out[i].resize(sends[i].bytes);
        if (sends[i].bytes && hipMemcpy(out[i].data(), sends[i].ptr_dev, sends[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    }
    std::atomic<bool> sent{true};
    std::vector<int> peers;
    for (int i = 0; i < ns; ++i) if (std::find(peers.begin(), peers.end(), sends[i].peer) == peers.end()) peers.push_back(sends[i].peer);
    std::vector<std::thread> txs;
    for (int peer : peers) txs.emplace_back([&, peer] { for (int i = 0; i < ns; ++i) if (sends[i].peer == peer && !wr_all(s->fd[peer], out[i].data(), out[i].size())) { sent = false; return; } });      // (list order per peer)
    bool got = true;
    for (int i = 0; i < nr && got; ++i) {
        if (recvs[i].peer < 0 || recvs[i].peer >= s->n || recvs[i].peer == s->rank) { got = false; break; }
        in[i].resize(recvs[i].bytes);
        got = rd_all(s->fd[recvs[i].peer], in[i].data(), in[i].size());
    }
    for (auto& t : txs) t.join();
    if (!sent || !got) return 1;
    for (int i = 0; i < nr; ++i) if (recvs[i].bytes && hipMemcpy(recvs[i].ptr_dev, in[i].data(), recvs[i].bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
}  // namespace

extern "C" {
int psgsdf_comm_unique_id(uint8_t id[128]) { return id ? comm_unique_id(id) : PSGSDF_ERR_ARG; }
static int attach(psgsdf_ctx* c, int rank, int n_ranks) {
    if (!c || rank < 0 || n_ranks < 1 || rank >= n_ranks) return PSGSDF_ERR_ARG;
    c->rank = rank; c->n_ranks = n_ranks; c->inited = false; c->have_volume = false;      // (the slab is chosen when the volume is uploaded)
    return PSGSDF_OK;
}
int psgsdf_comm_init(psgsdf_ctx* c, const uint8_t id[128], int rank, int n_ranks) {
    int rc = attach(c, rank, n_ranks); if (rc) return rc;
    if (!id) return fail(c, PSGSDF_ERR_ARG, "comm_init: the id of psgsdf_comm_unique_id is required");
    return comm_create_rccl(c, id, rank, n_ranks);
}
int psgsdf_comm_init_ext(psgsdf_ctx* c, const psgsdf_comm_ops* ops, int rank, int n_ranks) {
    int rc = attach(c, rank, n_ranks); if (rc) return rc;
    return comm_create_ext(c, ops, rank, n_ranks);
}
int psgsdf_comm_init_sockets(psgsdf_ctx* c, const int* peer_fd, int rank, int n_ranks) {
    int rc = attach(c, rank, n_ranks); if (rc) return rc;
    if (!peer_fd) return fail(c, PSGSDF_ERR_ARG, "comm_init_sockets: one connected stream socket per peer is required");
    auto sc = std::make_shared<SockComm>();
    sc->rank = rank; sc->n = n_ranks; sc->fd.assign(peer_fd, peer_fd + n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) continue;
        if (sc->fd[r] < 0) return fail(c, PSGSDF_ERR_ARG, "comm_init_sockets: no socket for rank %d", r);
        double t = 120.0; if (const char* e = getenv("PSGSDF_SOCKET_TIMEOUT_S")) t = atof(e);      // a peer that died must end the run, not hang it
        struct timeval tv; tv.tv_sec = (long)t; tv.tv_usec = (long)((t - (double)tv.tv_sec) * 1e6);
        if (setsockopt(sc->fd[r], SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv) != 0 || setsockopt(sc->fd[r], SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv) != 0)
            return fail(c, PSGSDF_ERR_COMM, "comm_init_sockets: the socket to rank %d takes no timeout (%s): a lost peer would hang this rank", r, strerror(errno));
    }
    psgsdf_comm_ops ops{sc.get(), &sock_allreduce, &sock_sendrecv};
    c->comm_keep = sc;
    return comm_create_ext(c, &ops, rank, n_ranks);
}
int psgsdf_comm_allreduce_host(psgsdf_ctx* c, double* buf, int n) {
    if (!c || !buf || n < 0) return PSGSDF_ERR_ARG;
    if (c->n_ranks <= 1 || n == 0) return PSGSDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> v(buf, buf + n);
    int rc = host_allreduce(c, v, "psgsdf_comm_allreduce_host"); if (rc) return rc;
    std::copy(v.begin(), v.end(), buf);
    return PSGSDF_OK;
}
int psgsdf_set_stream(psgsdf_ctx* c, void* hip_stream) {
    if (!c) return PSGSDF_ERR_ARG;
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream; c->own_stream = false;
    return PSGSDF_OK;
}
int psgsdf_mg_info(psgsdf_ctx* c, int32_t out[12]) {
    if (!c || !(c->inited || c->have_volume)) return PSGSDF_ERR_STATE;
    if (!c->inited) { for (int i = 0; i < 6; ++i) out[i] = 0; out[5] = c->F; }      // (a volume without a band yet: the planes only)
    else { out[0] = (int32_t)c->S_global; out[1] = c->band.S; out[2] = c->row0; out[3] = c->row1; out[4] = c->halo; out[5] = c->F; } out[6] = c->rank; out[7] = c->n_ranks;
    out[8] = c->need[0]; out[9] = c->need[1]; out[10] = c->z0; out[11] = c->z1;
    return PSGSDF_OK;
}
int psgsdf_comm_stats(psgsdf_ctx* c, int64_t* n_collectives) { if (!c || !n_collectives) return PSGSDF_ERR_ARG; *n_collectives = c->n_collectives; return PSGSDF_OK; }
}  // extern "C"
