// ps_optimizer.hpp — C++ mirror of the reference's optimiser interface over the C ABI (include/psgsdf.h).
//
// Same class / member names, argument meaning and return conventions as
//   ps_optimizer/Optimizer.h:24-182, PsOptimizer.h, LedOptimizer.h, OptimizerSettings.h:24-51
// so that main_ps.cpp:193-202,323-330 reads the same against this header.  The reference's Eigen / OpenCV
// types are replaced by plain structs (none of those libraries is a dependency of this project):
//   Eigen::Matrix4f  -> Mat4f   (row-major 16 floats)        cv::Mat (CV_32FC3, BGR) -> ImageRGB (float RGB)
//   VolumetricGradSdf -> VolumetricGradSdf below: the SoA form of its private tsdf_/vis_ arrays, which the
//   reference's optimisers reach through `friend` access (VolumetricGradSdf.h:25-42).
// All numerical work happens in libpsgsdf.so on the GPU; this header only marshals and writes the text files
// the reference writes from inside alternatingOptimize (optimizer_doc.txt, after_poses_opt_<k>.txt,
// *_pointcloud.ply).  Mesh extraction (marching cubes) is a SURVEY §8f "next" row and not part of it.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <atomic>
#include <thread>
#include <cstdint>
#include <fcntl.h>
#include <unistd.h>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/psgsdf.h"
#include "marching_cubes.hpp"

namespace psgsdf_host {

// Wall-clock per stage of a voxelPS run (`voxelPS --timing <file.json>`: VERDICT r04 item 4 -- the Gauss-Newton iterations are milliseconds, the
// decode / fusion / dump stages around them are what a user waits for).  Scopes nest: a stage's time includes the stages opened inside it.
struct StageClock {
    std::vector<std::pair<std::string, double>> acc; std::vector<long long> calls; std::mutex mu;      // (the background writer thread books its stages too)
    size_t slot(const std::string& n) { std::lock_guard<std::mutex> g(mu); for (size_t i = 0; i < acc.size(); ++i) if (acc[i].first == n) return i; acc.emplace_back(n, 0.0); calls.push_back(0); return acc.size() - 1; }
    struct Scope {
        StageClock& c; size_t i; std::chrono::steady_clock::time_point t0;
        Scope(StageClock& c_, const std::string& n) : c(c_), i(c_.slot(n)), t0(std::chrono::steady_clock::now()) {}
        ~Scope() { const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); std::lock_guard<std::mutex> g(c.mu); c.acc[i].second += dt; c.calls[i] += 1; }
    };
    static StageClock& get() { static StageClock c; return c; }
    double of(const std::string& n) { return acc[slot(n)].second; }
    bool write_json(const std::string& path, const std::string& extra = "") {
        std::ofstream f(path.c_str()); if (!f.is_open()) return false;
        f << "{\"stages_s\": {";
        for (size_t i = 0; i < acc.size(); ++i) f << (i ? ", " : "") << "\"" << acc[i].first << "\": " << acc[i].second;
        f << "}, \"calls\": {";
        for (size_t i = 0; i < acc.size(); ++i) f << (i ? ", " : "") << "\"" << acc[i].first << "\": " << calls[i];
        f << "}" << extra << "}\n";
        return true;
    }
};
#define PSG_STAGE_CAT2(a, b) a##b
#define PSG_STAGE_CAT(a, b) PSG_STAGE_CAT2(a, b)
#define PSG_STAGE(name) ::psgsdf_host::StageClock::Scope PSG_STAGE_CAT(psg_stage_, __LINE__)(::psgsdf_host::StageClock::get(), name)

using Mat4f = std::array<float, 16>;                 // row-major camera->world
struct Mat3f { float v[9]; };                        // row-major intrinsics
struct ImageRGB { int rows = 0, cols = 0; std::vector<float> data; };   // rows*cols*3, RGB in [0,1]

enum LossFunction { L2 = 0, CAUCHY = 1, HUBER = 2, TUKEY = 3, TRUNC_L2 = 4 };   // OptimizerSettings.h:9-16
enum ModelType { SH1, SH2, LED };                                               // OptimizerSettings.h:18-22

struct OptimizerSettings {                           // OptimizerSettings.h:24-51 (same defaults)
    int max_it = 100;
    float conv_threshold = 1e-4f;
    float damping = 1.0f;
    float lambda = 0.5f;
    float lambda_sq = 0.25f;
    float reg_weight_rho = 0.0f, reg_weight_n = 0.0f, reg_weight_l = 0.0f;
    int order = 1;
    bool upsample = false;
    ModelType model = SH1;
    LossFunction loss = CAUCHY;
};

// ---- the ASCII writers -----------------------------------------------------------------------------------------------------------------
// The reference prints every number through `ostream << float`, i.e. printf's %g with six significant digits, one std::endl (= a flush) per line;
// its files are tens of MB of that.  Same characters here, from snprintf into per-thread buffers (the lines of a file are formatted in parallel
// chunks and written in order), on a background thread, so that the optimisation goes on while a dump is being formatted.
inline bool& host_writers() { static bool v = false; return v; }
// voxelPS reads nothing of the refined volume after alternatingOptimize (main_ps.cpp:330-343 ends there): the executable skips the whole-volume download
// that mirrors the reference's in-place mutation of tSDF_ (a host that goes on using tSDF_ keeps it: the default)
inline bool& skip_sync_back() { static bool v = false; return v; }      // voxelPS --host-writers: round 4's path (dense download, host marching cubes, iostream): the cross-check
// printf's "%g" (six significant digits, trailing zeros stripped, scientific below 1e-4 and from 1e6) without printf: a dump is 3-5 million numbers and
// snprintf takes ~150 ns for each.  The float is exact in a double; scaled by an EXACT power of ten (10^0 .. 10^22) the product is off by at most one
// rounding (1.1e-16 relative, < 2e-10 at six digits), so the six digits are decided unless the value sits within 1e-6 of a rounding tie -- those, the
// non-finite values and the exponents outside the exact powers' reach go to snprintf.  `voxelPS --selftest-fmt N` compares with `ostream << float`.
inline int format_g6(float v, char* out) {
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    char* o = out;
    if (v == 0.0f) { if (std::signbit(v)) *o++ = '-'; *o++ = '0'; return (int)(o - out); }
    if (!std::isfinite(v)) return -1;
    const double a = std::fabs((double)v);
    int e2; std::frexp(a, &e2);
    int X = (int)std::floor((e2 - 1) * 0.30102999566398120);      // floor(log10(a)) or one less
    double x;
    for (int tries = 0;; ++tries) {
        const int sh = 5 - X;                                       // a * 10^sh in [1e5, 1e6)
        if (sh < -22 || sh > 22 || tries > 3) return -1;
        x = sh >= 0 ? a * p10[sh] : a / p10[-sh];
        if (x < 1e5) { --X; continue; }
        if (x >= 1e6) { ++X; continue; }
        break;
    }
    const double fl = std::floor(x), fr = x - fl;
    if (std::fabs(fr - 0.5) < 1e-6) return -1;                    // too close to a tie for this arithmetic to call
    long D = (long)fl + (fr > 0.5 ? 1 : 0);
    if (D >= 1000000) { D = 100000; ++X; }
    char d[6]; for (int k = 5; k >= 0; --k) { d[k] = (char)('0' + D % 10); D /= 10; }
    int nd = 6; while (nd > 1 && d[nd - 1] == '0') --nd;
    if (std::signbit(v)) *o++ = '-';
    if (X < -4 || X >= 6) {
        *o++ = d[0];
        if (nd > 1) { *o++ = '.'; for (int k = 1; k < nd; ++k) *o++ = d[k]; }
        *o++ = 'e'; *o++ = X < 0 ? '-' : '+';
        const int ax = X < 0 ? -X : X;
        if (ax >= 100) *o++ = (char)('0' + ax / 100);
        *o++ = (char)('0' + (ax / 10) % 10); *o++ = (char)('0' + ax % 10);
    } else if (X >= 0) {
        for (int k = 0; k <= X; ++k) *o++ = k < nd ? d[k] : '0';
        if (nd > X + 1) { *o++ = '.'; for (int k = X + 1; k < nd; ++k) *o++ = d[k]; }
    } else {
        *o++ = '0'; *o++ = '.';
        for (int k = 0; k < -X - 1; ++k) *o++ = '0';
        for (int k = 0; k < nd; ++k) *o++ = d[k];
    }
    return (int)(o - out);
}
struct TextOut {
    std::string s;
    void f(float v) { char t[40]; int n = format_g6(v, t); if (n < 0) n = snprintf(t, sizeof t, "%g", (double)v); s.append(t, (size_t)n); }
    void i(long long v) {
        char t[24]; int n = 24; unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { t[--n] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) t[--n] = '-';
        s.append(t + n, (size_t)(24 - n));
    }
    void c(char ch) { s.push_back(ch); }
};
// n lines, line(out, i) appends line i; formatted on up to 16 threads, in order
template <class Fn>
inline std::vector<TextOut> format_lines(size_t n, size_t approx_line_bytes, Fn line) {
    unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < 50000) T = 1;
    std::vector<TextOut> parts(T);
    auto work = [&](unsigned t) { const size_t a = n * t / T, b = n * (t + 1) / T; parts[t].s.reserve((b - a) * approx_line_bytes); for (size_t q = a; q < b; ++q) line(parts[t], q); };
    if (T == 1) work(0);
    else { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    return parts;
}
template <class Fn>
inline bool write_lines(FILE* f, size_t n, size_t approx_line_bytes, Fn line) {
    for (auto& p : format_lines(n, approx_line_bytes, line)) if (!p.s.empty() && fwrite(p.s.data(), 1, p.s.size(), f) != p.s.size()) return false;
    return true;
}
inline void mesh_vertex_line(TextOut& o, const float* xyz, const uint8_t* rgb, size_t i) { o.f(xyz[3 * i]); o.c(' '); o.f(xyz[3 * i + 1]); o.c(' '); o.f(xyz[3 * i + 2]); o.c(' '); o.i(rgb[3 * i]); o.c(' '); o.i(rgb[3 * i + 1]); o.c(' '); o.i(rgb[3 * i + 2]); o.c('\n'); }
inline void mesh_face_line(TextOut& o, size_t q) { o.c('3'); o.c(' '); o.i((long long)(3 * q)); o.c(' '); o.i((long long)(3 * q + 1)); o.c(' '); o.i((long long)(3 * q + 2)); o.c('\n'); }
inline void pointcloud_line(TextOut& o, const float* pn, const int32_t* col, size_t i) { for (int k = 0; k < 6; ++k) { o.f(pn[6 * i + k]); o.c(' '); } o.i(col[3 * i]); o.c(' '); o.i(col[3 * i + 1]); o.c(' '); o.i(col[3 * i + 2]); o.c('\n'); }
inline std::string mesh_header(size_t nv) {
    char h[400]; const int n = snprintf(h, sizeof h, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
               "element face %d\nproperty list uchar int vertex_indices\nend_header\n", nv, (int)(nv / 3));
    return std::string(h, (size_t)n);
}
inline std::string pointcloud_header(size_t n) {
    char h[400]; const int k = snprintf(h, sizeof h, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
               "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", n);
    return std::string(h, (size_t)k);
}
inline std::string sdf_header(const int lo[3], const int dim[3], float vs) {
    TextOut h; h.i(dim[0]); h.c(' '); h.i(dim[1]); h.c(' '); h.i(dim[2]); h.c('\n'); h.f(lo[0] * vs); h.c(' '); h.f(lo[1] * vs); h.c(' '); h.f(lo[2] * vs); h.c('\n'); h.f(vs); h.c('\n');
    return h.s;
}
// MarchingCubes::savePly (third/mesh/MarchingCubes.cpp:659-699) from non-indexed vertices: 3 per face
inline bool write_mesh_ply(const std::string& file, const float* xyz, const uint8_t* rgb, size_t nv) {
    if (nv == 0) return false;
    FILE* f = fopen(file.c_str(), "wb"); if (!f) return false;
    const std::string h = mesh_header(nv);
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    ok = ok && write_lines(f, nv, 40, [&](TextOut& o, size_t i) { mesh_vertex_line(o, xyz, rgb, i); });
    ok = ok && write_lines(f, nv / 3, 24, [&](TextOut& o, size_t q) { mesh_face_line(o, q); });
    return fclose(f) == 0 && ok;
}
// save_pointcloud / extract_pc (OptimizerAux.cpp:456-511, VolumetricGradSdf.cpp:320-376): x y z nx ny nz r g b
inline bool write_pointcloud_ply(const std::string& file, const float* pn, const int32_t* col, size_t n) {
    FILE* f = fopen(file.c_str(), "wb"); if (!f) return false;
    const std::string h = pointcloud_header(n);
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    ok = ok && write_lines(f, n, 80, [&](TextOut& o, size_t i) { pointcloud_line(o, pn, col, i); });
    return fclose(f) == 0 && ok;
}
// saveSDF (OptimizerAux.cpp:513-577): dims, the box's first voxel in metres, the voxel size, then -dist, x fastest
inline bool write_sdf_block(const std::string& file, const int lo[3], const int dim[3], float vs, const float* v) {
    FILE* f = fopen(file.c_str(), "wb"); if (!f) return false;
    const std::string h = sdf_header(lo, dim, vs);
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    ok = ok && write_lines(f, (size_t)dim[0] * dim[1] * dim[2], 12, [&](TextOut& o, size_t i) { o.f(v[i]); o.c('\n'); });
    return fclose(f) == 0 && ok;
}
// One background thread that formats and writes the dumps in the order they were issued; drain() before anything that must see the files.
class DumpQueue {
    std::thread th_; std::mutex m_; std::condition_variable cv_, idle_; std::deque<std::function<void()>> q_; bool stop_ = false, busy_ = false;
    void run() {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return stop_ || !q_.empty(); }); if (q_.empty()) return; job = std::move(q_.front()); q_.pop_front(); busy_ = true; }
            { PSG_STAGE("background: format + write the dump files"); job(); }
            { std::lock_guard<std::mutex> l(m_); busy_ = false; }
            idle_.notify_all();
        }
    }
    std::atomic<int> failures_{0};
public:
    static DumpQueue& get() { static DumpQueue q; return q; }
    void report_failure() { failures_.fetch_add(1); }      // a job could not open / write its file: surfaced by failures() after drain() (voxelPS exits non-zero)
    int failures() const { return failures_.load(); }
    void push(std::function<void()> job) {
        if (host_writers()) { job(); return; }
        std::lock_guard<std::mutex> l(m_);
        if (!th_.joinable()) th_ = std::thread([this] { run(); });
        q_.push_back(std::move(job)); cv_.notify_one();
    }
    void drain() { PSG_STAGE("wait for the background writer"); std::unique_lock<std::mutex> l(m_); idle_.wait(l, [&] { return q_.empty() && !busy_; }); }
    ~DumpQueue() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); if (th_.joinable()) th_.join(); }
};

// ---- one run on N devices (`voxelPS --gpus N`: one process per GPU, the volume cut into z-slabs; no reference counterpart) ---------------------------
// main() fills this in before anything else.  Every rank runs the same program on the same inputs (decodes the frames, selects the keyframes); the
// engine calls are collective.  Rank 0 narrates and writes the small text files; the big ones are written by ALL ranks: each formats the lines of
// its share (psgsdf_extract_* on a multi-rank context) and writes them into their place in the one file -- the places come from an exclusive scan of
// the byte counts over the ranks (psgsdf_comm_allreduce_host).  The files are the single-process files line for line, up to what the slabs'
// rank-order sums do to the last digit of a value.
struct RankInfo { int rank = 0, n = 1, device = 0; bool sockets = false; std::vector<int> fds; std::vector<uint8_t> id; };
inline RankInfo& rank_info() { static RankInfo r; return r; }
inline bool multi_rank() { return rank_info().n > 1; }
inline bool lead_rank() { return rank_info().rank == 0; }
// k numbers of every rank -> the sums over the ranks in front of this one, and over all
// `local_ok` = false: this rank failed in front of the collective.  It STILL enters it (a rank that returned early left the others blocked in the
// all-reduce: ADVICE r05) and its flag travels as one more summed element, so that every rank returns false together.
inline bool scan_over_ranks(psgsdf_ctx* ctx, const std::vector<double>& mine, std::vector<double>& before, std::vector<double>& total, bool local_ok = true) {
    const RankInfo& ri = rank_info(); const size_t k = mine.size(), k1 = k + 1;
    std::vector<double> all(k1 * (size_t)ri.n, 0.0);
    for (size_t i = 0; i < k; ++i) all[k1 * (size_t)ri.rank + i] = local_ok ? mine[i] : 0.0;
    all[k1 * (size_t)ri.rank + k] = local_ok ? 0.0 : 1.0;
    if (psgsdf_comm_allreduce_host(ctx, all.data(), (int)all.size())) return false;
    before.assign(k, 0.0); total.assign(k, 0.0);
    double failed = 0.0;
    for (int r = 0; r < ri.n; ++r) { for (size_t i = 0; i < k; ++i) { if (r < ri.rank) before[i] += all[k1 * (size_t)r + i]; total[i] += all[k1 * (size_t)r + i]; } failed += all[k1 * (size_t)r + k]; }
    return failed == 0.0;
}
inline size_t bytes_of(const std::vector<TextOut>& parts) { size_t b = 0; for (auto& p : parts) b += p.s.size(); return b; }
struct FilePiece { long long offset; std::shared_ptr<std::vector<TextOut>> parts; };
// this rank's pieces of a file all ranks write: rank 0 also owns the header and the file's final length
inline void write_shared_file(const std::string& file, const std::string& header, long long total_bytes, std::vector<FilePiece> pieces) {
    DumpQueue::get().push([=] {
        const int fd = open(file.c_str(), O_CREAT | O_WRONLY, 0644);      // (no O_TRUNC: the other ranks write their pieces into the same file at the same time; the lead rank sets the final length)
        if (fd < 0) { std::cerr << "couldn't open " << file << std::endl; DumpQueue::get().report_failure(); return; }
        auto put = [&](const char* p, size_t n, long long at) { while (n) { const ssize_t k = pwrite(fd, p, n, (off_t)at); if (k <= 0) { std::cerr << "couldn't write " << file << std::endl; return false; } p += k; n -= (size_t)k; at += k; } return true; };
        bool ok = true;
        if (lead_rank()) ok = put(header.data(), header.size(), 0) && ftruncate(fd, (off_t)total_bytes) == 0;
        for (auto& pc : pieces) { long long at = pc.offset; for (auto& t : *pc.parts) { if (ok && !t.s.empty()) ok = put(t.s.data(), t.s.size(), at); at += (long long)t.s.size(); } }
        if (close(fd) != 0) ok = false;
        if (!ok) DumpQueue::get().report_failure();      // (this rank's piece is missing from a file the other ranks completed: the run reports it, voxelps_main.cpp)
    });
}

struct DepthImage;
// writers shared by VolumetricGradSdf (init_* files) and Optimizer (refined files): cropped box of |d| <= sqrt(3) vs,
// -dist, grid-local coordinates (VolumetricGradSdf.cpp:234-318,379-442, OptimizerAux.cpp:278-363,513-577)
struct CropBox { int lo[3], hi[3]; bool any; };
inline CropBox crop_box(const int dim[3], const std::vector<float>& dist, float vs) {
    CropBox b{{1 << 30, 1 << 30, 1 << 30}, {-(1 << 30), -(1 << 30), -(1 << 30)}, false};
    size_t lin = 0;
    for (int k = 0; k < dim[2]; ++k) for (int j = 0; j < dim[1]; ++j) for (int i = 0; i < dim[0]; ++i, ++lin) {
        if (std::fabs(dist[lin]) > std::sqrt(3) * vs) continue;
        int idx[3] = {i, j, k};
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], idx[a]); b.hi[a] = std::max(b.hi[a], idx[a]); }
        b.any = true;
    }
    return b;
}
inline bool write_mesh(const std::string& file, const int dim[3], float vs, const std::vector<float>& dist, const std::vector<float>& weight, const std::vector<float>& rgb) {
    CropBox b = crop_box(dim, dist, vs); if (!b.any) return false;
    const size_t n = (size_t)dim[0] * dim[1] * dim[2];
    int d[3] = {b.hi[0] - b.lo[0] + 1, b.hi[1] - b.lo[1] + 1, b.hi[2] - b.lo[2] + 1};
    const size_t nv = (size_t)d[0] * d[1] * d[2];
    std::vector<float> t(nv), w(nv); std::vector<unsigned char> r(nv), g(nv), bl(nv);
    size_t pos = 0;
    for (int k = b.lo[2]; k <= b.hi[2]; ++k) for (int j = b.lo[1]; j <= b.hi[1]; ++j) for (int i = b.lo[0]; i <= b.hi[0]; ++i, ++pos) {
        size_t lin = (size_t)i + (size_t)j * dim[0] + (size_t)k * dim[0] * dim[1];
        t[pos] = -dist[lin]; w[pos] = weight[lin];
        r[pos] = (unsigned char)int(255 * rgb[lin]); g[pos] = (unsigned char)int(255 * rgb[n + lin]); bl[pos] = (unsigned char)int(255 * rgb[2 * n + lin]);
    }
    float size[3] = {vs * d[0], vs * d[1], vs * d[2]}, org[3] = {-vs * b.lo[0], -vs * b.lo[1], -vs * b.lo[2]};
    MarchingCubes mc(d, size, org);
    mc.computeIsoSurface(t.data(), w.data(), r.data(), g.data(), bl.data());
    return mc.savePly(file);
}
inline bool write_sdf(const std::string& file, const int dim[3], float vs, const std::vector<float>& dist) {
    CropBox b = crop_box(dim, dist, vs); if (!b.any) return false;
    std::ofstream f(file.c_str()); if (!f.is_open()) return false;
    f << b.hi[0] - b.lo[0] + 1 << " " << b.hi[1] - b.lo[1] + 1 << " " << b.hi[2] - b.lo[2] + 1 << "\n";
    f << b.lo[0] * vs << " " << b.lo[1] * vs << " " << b.lo[2] * vs << "\n" << vs << "\n";
    for (int k = b.lo[2]; k <= b.hi[2]; ++k) for (int j = b.lo[1]; j <= b.hi[1]; ++j) for (int i = b.lo[0]; i <= b.hi[0]; ++i)
        f << -dist[(size_t)i + (size_t)j * dim[0] + (size_t)k * dim[0] * dim[1]] << "\n";
    return true;
}

// SoA image of VolumetricGradSdf's state (x-fastest, VoxelGrid.h:79-82).  Two modes: host arrays handed to the optimiser
// (voxelps_scene), or -- when created with `attach` -- a volume that lives on the device from the first frame on
// (voxelPS: VolumetricGradSdf::update and the tracker run there too).
struct VolumetricGradSdf {
    int grid_dim_[3] = {0, 0, 0};
    float voxel_size_ = 0, T_ = 0;
    float shift_[3] = {0, 0, 0};
    float z_min_ = 0.5f, z_max_ = 10.0f;             // Sdf.h:40-41
    size_t counter_ = 0;
    std::vector<float> dist, grad, weight, rgb;      // N, 3N (x|y|z), N, 3N (r|g|b)
    std::vector<uint64_t> vis; int vis_words = 1;    // N*vis_words, bit c = seen by integrated frame c
    psgsdf_ctx* ctx = nullptr;                       // device-resident mode (owned)
    size_t num_voxels() const { return (size_t)grid_dim_[0] * grid_dim_[1] * grid_dim_[2]; }
    ~VolumetricGradSdf() { if (ctx) psgsdf_destroy(ctx); }

    // VolumetricGradSdf(grid_dim, voxel_size, shift, T) + init() on the device (main_ps.cpp:183)
    bool attach(const int dim[3], float voxel_size, const float shift[3], float T, const Mat3f& K, const psgsdf_settings& s, int max_frames) {
        for (int a = 0; a < 3; ++a) { grid_dim_[a] = dim[a]; shift_[a] = shift[a]; }
        voxel_size_ = voxel_size; T_ = T;
        psgsdf_grid_desc g{}; for (int a = 0; a < 3; ++a) { g.dim[a] = dim[a]; g.shift[a] = shift[a]; } g.voxel_size = voxel_size; g.truncation = T;
        const RankInfo& ri = rank_info();
        if (psgsdf_create(&g, K.v, &s, ri.device, &ctx)) { ctx = nullptr; return false; }
        if (ri.n > 1) {      // one process per GPU: RCCL over xGMI, or the node-local socket transport (ranks that share a device; no RCCL)
            const int rc = ri.sockets ? psgsdf_comm_init_sockets(ctx, ri.fds.data(), ri.rank, ri.n) : psgsdf_comm_init(ctx, ri.id.data(), ri.rank, ri.n);
            if (rc) { std::cerr << "rank " << ri.rank << ": no communicator: " << last_error() << std::endl; return false; }
        }
        std::cout << "Number of voxels: " << num_voxels() << std::endl;
        return psgsdf_volume_init(ctx, max_frames) == 0;
    }
    const char* last_error() const { const char* e = ctx ? psgsdf_last_error(ctx) : "no device volume"; return e ? e : ""; }
    void set_zmin(float z) { z_min_ = z; }
    void set_zmax(float z) { z_max_ = z; }
    void increase_counter() { ++counter_; }
    // VolumetricGradSdf::update (VolumetricGradSdf.cpp:51-138): FALS normals + fusion, both on the device
    bool update(const ImageRGB& color, const std::vector<float>& depth, const Mat4f& pose) {
        PSG_STAGE("fuse: FALS normals + integration (device, incl. transfers)");
        if (host_writers()) {      // (round 4's path: the normals come back to the host and go up again)
            std::vector<float> nrm((size_t)3 * color.rows * color.cols);
            if (psgsdf_estimate_normals(ctx, depth.data(), color.cols, color.rows, nrm.data())) return false;
            return psgsdf_integrate_frame(ctx, color.data.data(), depth.data(), nrm.data(), color.cols, color.rows, pose.data(), (int)counter_, z_min_, z_max_) == 0;
        }
        return psgsdf_integrate_frame(ctx, color.data.data(), depth.data(), nullptr, color.cols, color.rows, pose.data(), (int)counter_, z_min_, z_max_) == 0;      // NULL: FALS normals on the device
    }
    bool sync_host() {
        const size_t n = num_voxels();
        dist.resize(n); grad.resize(3 * n); weight.resize(n); rgb.resize(3 * n);
        return psgsdf_download_volume(ctx, dist.data(), grad.data(), weight.data(), rgb.data(), nullptr) == 0;
    }
    bool extract_mesh(const std::string& filename) {
        PSG_STAGE("dump: mesh (marching cubes + PLY)");
        if (ctx && !host_writers()) return device_mesh(ctx, filename);
        return sync_host() && write_mesh(filename, grid_dim_, voxel_size_, dist, weight, rgb);
    }
    bool saveSDF(const std::string& filename) {
        PSG_STAGE("dump: sdf");
        if (ctx && !host_writers()) return device_sdf(ctx, filename, voxel_size_);
        return sync_host() && write_sdf(filename, grid_dim_, voxel_size_, dist);
    }
    // the device-side extraction (include/psgsdf.h psgsdf_extract_*): compact arrays come back, a copy of them goes to the background writer
    static bool device_mesh(psgsdf_ctx* ctx, const std::string& file) {
        const float* xyz = nullptr; const uint8_t* rgb = nullptr; int64_t nv = 0;
        const bool ok = psgsdf_extract_mesh(ctx, &xyz, &rgb, &nv) == 0;
        if (!ok) nv = 0;
        if (multi_rank()) {
            std::vector<double> before, total;
            if (!scan_over_ranks(ctx, {(double)nv}, before, total, ok) || total[0] == 0) return false;
            auto V = std::make_shared<std::vector<TextOut>>(format_lines((size_t)nv, 40, [&](TextOut& o, size_t i) { mesh_vertex_line(o, xyz, rgb, i); }));
            const size_t face0 = (size_t)before[0] / 3;      // (faces are numbered through the whole file)
            auto F = std::make_shared<std::vector<TextOut>>(format_lines((size_t)nv / 3, 24, [&](TextOut& o, size_t q) { mesh_face_line(o, face0 + q); }));
            const std::string h = mesh_header((size_t)total[0]);
            std::vector<double> bb, bt;
            if (!scan_over_ranks(ctx, {(double)bytes_of(*V), (double)bytes_of(*F)}, bb, bt)) return false;
            write_shared_file(file, h, (long long)(h.size() + bt[0] + bt[1]), {{(long long)(h.size() + bb[0]), V}, {(long long)(h.size() + bt[0] + bb[1]), F}});
            return true;
        }
        if (!ok || nv == 0) return false;
        auto vx = std::make_shared<std::vector<float>>(xyz, xyz + 3 * nv); auto vc = std::make_shared<std::vector<uint8_t>>(rgb, rgb + 3 * nv);
        DumpQueue::get().push([=] { if (!write_mesh_ply(file, vx->data(), vc->data(), (size_t)nv)) std::cout << "couldn't save mesh " << file << std::endl; });
        return true;
    }
    static bool device_pointcloud(psgsdf_ctx* ctx, int which, const std::string& file) {
        const float* pn = nullptr; const int32_t* col = nullptr; int64_t n = 0;
        const bool ok = psgsdf_extract_pointcloud(ctx, which, &pn, &col, &n) == 0;
        if (!ok) n = 0;
        if (multi_rank()) {
            auto L = std::make_shared<std::vector<TextOut>>(format_lines((size_t)n, 80, [&](TextOut& o, size_t i) { pointcloud_line(o, pn, col, i); }));
            std::vector<double> before, total;
            if (!scan_over_ranks(ctx, {(double)n, (double)bytes_of(*L)}, before, total, ok)) return false;
            const std::string h = pointcloud_header((size_t)total[0]);
            write_shared_file(file, h, (long long)(h.size() + total[1]), {{(long long)(h.size() + before[1]), L}});
            return true;
        }
        if (!ok) return false;
        auto vp = std::make_shared<std::vector<float>>(pn, pn + 6 * n); auto vc = std::make_shared<std::vector<int32_t>>(col, col + 3 * n);
        DumpQueue::get().push([=] { if (!write_pointcloud_ply(file, vp->data(), vc->data(), (size_t)n)) std::cout << " can't save point cloud!" << std::endl; });
        return true;
    }
    static bool device_sdf(psgsdf_ctx* ctx, const std::string& file, float vs) {
        int32_t lo[3], dim[3]; const float* v = nullptr;
        bool ok = psgsdf_extract_sdf(ctx, lo, dim, &v) == 0;
        if (multi_rank()) {
            int32_t mi[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && psgsdf_mg_info(ctx, mi)) ok = false;
            if (!ok) dim[0] = dim[1] = dim[2] = lo[0] = lo[1] = lo[2] = 0;
            const size_t cnt = (size_t)dim[0] * dim[1] * (size_t)std::max(0, std::min(lo[2] + dim[2], mi[11]) - std::max(lo[2], mi[10]));      // the planes of the box this rank owns
            auto L = std::make_shared<std::vector<TextOut>>(format_lines(cnt, 12, [&](TextOut& o, size_t i) { o.f(v[i]); o.c('\n'); }));
            std::vector<double> before, total;
            if (!scan_over_ranks(ctx, {(double)bytes_of(*L)}, before, total, ok)) return false;
            if (dim[0] == 0) return false;      // (the crop box is the whole volume's: empty on every rank alike)
            const int l3[3] = {lo[0], lo[1], lo[2]}, d3[3] = {dim[0], dim[1], dim[2]};
            const std::string h = sdf_header(l3, d3, vs);
            write_shared_file(file, h, (long long)(h.size() + total[0]), {{(long long)(h.size() + before[0]), L}});
            return true;
        }
        if (!ok || !v) return false;
        auto vv = std::make_shared<std::vector<float>>(v, v + (size_t)dim[0] * dim[1] * dim[2]);
        const std::array<int, 3> l{lo[0], lo[1], lo[2]}, d{dim[0], dim[1], dim[2]};
        DumpQueue::get().push([=] { write_sdf_block(file, l.data(), d.data(), vs, vv->data()); });
        return true;
    }
    // extract_pc, VolumetricGradSdf.cpp:320-376: x y z nx ny nz r g b of every voxel with |d| < sqrt(3) vs and weight > 0
    bool extract_pc(const std::string& filename) {
        PSG_STAGE("dump: point cloud");
        if (ctx && !host_writers()) return device_pointcloud(ctx, 1, filename);
        if (!sync_host()) return false;
        const size_t n = num_voxels(); std::vector<size_t> sel;
        for (size_t lin = 0; lin < n; ++lin) if (weight[lin] > 0 && std::fabs(dist[lin]) < std::sqrt(3) * voxel_size_) sel.push_back(lin);
        std::ofstream ply(filename.c_str()); if (!ply.is_open()) return false;
        ply << "ply\nformat ascii 1.0\nelement vertex " << sel.size() << "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
            << "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header" << std::endl;
        const int nx = grid_dim_[0], nxy = grid_dim_[0] * grid_dim_[1];
        for (size_t lin : sel) {
            int k = (int)(lin / nxy), rest = (int)(lin - (size_t)k * nxy), j = rest / nx, i = rest - j * nx;
            float g[3] = {grad[lin], grad[n + lin], grad[2 * n + lin]}; float z = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
            if (z > 0) { float s = std::sqrt(z); g[0] /= s; g[1] /= s; g[2] /= s; }
            ply << voxel_size_ * i - dist[lin] * g[0] << " " << voxel_size_ * j - dist[lin] * g[1] << " " << voxel_size_ * k - dist[lin] * g[2] << " " << g[0] << " " << g[1] << " " << g[2] << " "
                << int(255 * rgb[lin]) << " " << int(255 * rgb[n + lin]) << " " << int(255 * rgb[2 * n + lin]) << std::endl;
        }
        return true;
    }
};

// RigidPointOptimizer (sdf_tracker/RigidPointOptimizer.{h,cpp}, RigidOptimizer.h:41-47): frame-to-model tracking on the device volume
class RigidPointOptimizer {
    VolumetricGradSdf* tSDF_;
    int num_iterations_ = 50; float conv_threshold_ = 1e-3f, damping_ = 1.0f;
    Mat4f pose_ = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
public:
    explicit RigidPointOptimizer(VolumetricGradSdf* tSDF) : tSDF_(tSDF) {}
    void set_pose(const Mat4f& p) { pose_ = p; }
    Mat4f pose() const { return pose_; }
    bool optimize(const std::vector<float>& depth, int cols, int rows) {
        PSG_STAGE("track: frame-to-model (device)");
        int iters = 0, conv = 0;
        if (psgsdf_track(tSDF_->ctx, depth.data(), cols, rows, pose_.data(), tSDF_->z_min_, tSDF_->z_max_, num_iterations_, conv_threshold_, damping_, &iters, &conv)) return false;
        if (conv) std::cout << "... Convergence after " << iters << " iterations!" << std::endl;
        return conv != 0;
    }
};

class Optimizer {
protected:
    VolumetricGradSdf* tSDF_;
    float voxel_size_;
    Mat3f K_;
    std::string save_path_;
    OptimizerSettings* settings_;
    std::vector<int> frame_idx_;
    std::vector<std::shared_ptr<ImageRGB>> images_;
    std::vector<Mat4f> poses_;
    std::vector<std::string> key_stamps_;
    psgsdf_ctx* ctx_ = nullptr; bool borrowed_ = false;
    size_t num_frames_ = 0, num_voxels_ = 0;
    std::ofstream doc_;

    bool fail(const char* what, int rc) {
        std::cerr << what << " failed (" << rc << "): " << (ctx_ ? psgsdf_last_error(ctx_) : "") << std::endl;
        return false;
    }
    static int on_iter_trampoline(void* user, int iter_done, const psgsdf_iter_stats* rec) { return static_cast<Optimizer*>(user)->on_iter(iter_done, rec); }

    // getTotalEnergy's log line, OptimizerAux.cpp:259-269
    void log_energy(double E, const psgsdf_iter_stats* r, bool before_dist = false) {
        // (the blocks in front of the distance block are logged with the regulariser energies of the previous iteration: E_n / E_l only change at PsOptimizer.cpp:342-343)
        double en = r->reg_weight_n * (before_dist ? r->e_n_in : r->e_n), el = r->reg_weight_l * (before_dist ? r->e_l_in : r->e_l), er = settings_->reg_weight_rho * r->e_r;
        for (std::ostream* o : {static_cast<std::ostream*>(&std::cout), static_cast<std::ostream*>(&doc_)})
            if (o == &std::cout || doc_.is_open())
                *o << "PS energy: " << E << "\t normal reg energy: " << en << "\t laplacian reg energy: " << el << "\t rho reg energy: " << er
                   << "\t total energy: " << (float)(E + en + el + er) << std::endl;
    }
    // the per-iteration narration of PsOptimizer.cpp:304-366: needs the record only, so it runs as the engine's PASSIVE record observer
    // (psgsdf_set_record_observer) and the loop may already be working on the next iteration while it prints
    int narrate(int iter_done, const psgsdf_iter_stats* r) {
        const int iter = iter_done - 1;
        const bool led = settings_->model == LED;
        const char* names[4] = {"albedo", "light", "distance", "pose"};
        const int order_sh[4] = {0, 1, 2, 3}, order_led[4] = {1, 0, 2, 3};
        for (int q = 0; q < 4; ++q) {
            int s = led ? order_led[q] : order_sh[q];
            if (std::isnan(r->e_after[s])) continue;
            std::cout << "===> [" << iter << "]: after " << names[s] << " optimization: ";
            if (doc_.is_open()) doc_ << "===> [" << iter << "]: after " << names[s] << " optimization: \n";
            log_energy(r->e_after[s], r, s < 2);
        }
        std::cout << "===> [" << iter << "]: relative diff " << r->rel_diff << std::endl;
        if (doc_.is_open()) doc_ << "===> [" << iter << "]: relative diff " << r->rel_diff << "\n";
        return 0;
    }
    static int narrate_trampoline(void* user, int iter_done, const psgsdf_iter_stats* rec) { return static_cast<Optimizer*>(user)->narrate(iter_done, rec); }
    // the periodic dumps of PsOptimizer.cpp:397-398,419-423 read the STATE of the iteration: the exact-state callback, due every 3rd iteration
    // (psgsdf_set_on_iter_period(3)) and after the 2x refinement; in between the loop closes its iterations speculatively
    int on_iter(int iter_done, const psgsdf_iter_stats* r) {
        const int iter = iter_done - 1;
        if (r->upsampled) {   // PsOptimizer.cpp:397-405
            save_pointcloud("upsample_after_" + std::to_string(iter));
            extract_mesh("upsample_after_" + std::to_string(iter));
            // the log line behind the refinement: the iteration's last PS energy and Eikonal term with the Laplacian energy of the REFINED grid under
            // its re-normalised weight (PsOptimizer.cpp:400-405: E_l = getLaplacianEnergy(); reg_weight_l *= E / E_l; getTotalEnergy(...))
            double e4[4] = {0, 0, 0, 0}; psgsdf_info info{};
            if (!psgsdf_energy(ctx_, e4) && !psgsdf_get_info(ctx_, &info)) {
                double E = 0; for (int q = 0; q < 4; ++q) if (!std::isnan(r->e_after[q])) E = r->e_after[q];      // (slot order albedo, light, dist, pose: the last enabled block is the last one run)
                psgsdf_iter_stats t = *r; t.e_l = e4[2]; t.reg_weight_l = info.reg_weight_l;
                std::cout << "===> [" << iter << "]: after pose optimization: ";
                if (doc_.is_open()) doc_ << "===> [" << iter << "]: after pose optimization: \n";
                log_energy(E, &t);
            }
        }
        if (iter_done % 3 == 0) {
            savePoses("after_poses_opt_" + std::to_string(iter_done));
            save_pointcloud("after_iter_" + std::to_string(iter_done));
            extract_mesh("after_iter_" + std::to_string(iter_done));
        }
        return 0;
    }

public:
    Optimizer(VolumetricGradSdf* tSDF, const float voxel_size, const Mat3f& K, std::string save_path, OptimizerSettings* settings)
        : tSDF_(tSDF), voxel_size_(voxel_size), K_(K), save_path_(save_path), settings_(settings) {}
    virtual ~Optimizer() { if (ctx_ && !borrowed_) psgsdf_destroy(ctx_); }

    void setImages(std::vector<std::shared_ptr<ImageRGB>> images) { images_ = images; }          // Optimizer.h:137-140
    void setPoses(std::vector<Mat4f>& pose) { poses_ = pose; }                                    // Optimizer.h:142-145
    void setKeyframes(std::vector<int>& keyframes) { frame_idx_ = keyframes; }                    // Optimizer.h:147-150
    void setKeytimestamps(std::vector<std::string>& keystamps) { key_stamps_ = keystamps; }      // Optimizer.h:151-154

    // PsOptimizer::init / LedOptimizer::init (PsOptimizer.cpp:25-42): with zero keyframes (the constructor-time
    // call of the reference) this is a no-op; with keyframes it creates the device context and uploads everything.
    virtual void init() {
        PSG_STAGE("optimiser init: keyframe upload + band");
        num_frames_ = frame_idx_.size();
        num_voxels_ = tSDF_->num_voxels();
        if (num_frames_ == 0) return;
        if (tSDF_->ctx) {   // device-resident volume: the fused state is already there
            ctx_ = tSDF_->ctx; borrowed_ = true;
            if (multi_rank()) { const int rb = psgsdf_rebalance_slabs(ctx_); if (rb) { fail("psgsdf_rebalance_slabs", rb); return; } }      // fused in slabs of equal height, optimised in slabs of equal band count
            const int W = images_[0]->cols, H = images_[0]->rows;
            std::vector<float> P(num_frames_ * 16); std::vector<const float*> img(num_frames_);
            for (size_t f = 0; f < num_frames_; ++f) {
                if (images_[f]->cols != W || images_[f]->rows != H) { std::cerr << "keyframe " << f << " has another size" << std::endl; return; }
                img[f] = images_[f]->data.data();      // (one allocation per image, like the reference's std::vector<cv::Mat>: no gather)
                std::copy(poses_[f].begin(), poses_[f].end(), P.begin() + f * 16);
            }
            int rc = psgsdf_set_keyframes_frames(ctx_, (int)num_frames_, frame_idx_.data(), img.data(), W, H, P.data());
            if (rc) { fail("psgsdf_set_keyframes", rc); return; }
            rc = psgsdf_init(ctx_);
            if (rc) fail("psgsdf_init", rc);
            return;
        }
        if (ctx_) { psgsdf_destroy(ctx_); ctx_ = nullptr; }
        psgsdf_grid_desc g{};
        for (int a = 0; a < 3; ++a) { g.dim[a] = tSDF_->grid_dim_[a]; g.shift[a] = tSDF_->shift_[a]; }
        g.voxel_size = voxel_size_; g.truncation = tSDF_->T_;
        psgsdf_settings s{};
        s.model = settings_->model == LED ? PSGSDF_LED : (settings_->model == SH2 ? PSGSDF_SH2 : PSGSDF_SH1);
        s.loss = (int)settings_->loss; s.lambda = settings_->lambda; s.damping = settings_->damping;
        s.reg_weight_rho = settings_->reg_weight_rho; s.reg_weight_n = settings_->reg_weight_n; s.reg_weight_l = settings_->reg_weight_l;
        s.max_it = settings_->max_it; s.conv_threshold = settings_->conv_threshold; s.upsample = settings_->upsample ? 1 : 0;
        s.ref_quirks = 1; s.cg_max_it = 0;
        int rc = psgsdf_create(&g, K_.v, &s, 0, &ctx_);
        if (rc) { fail("psgsdf_create", rc); ctx_ = nullptr; return; }
        rc = psgsdf_upload_volume(ctx_, tSDF_->dist.data(), tSDF_->grad.data(), tSDF_->weight.data(), tSDF_->rgb.data(), tSDF_->vis.data(), tSDF_->vis_words);
        if (rc) { fail("psgsdf_upload_volume", rc); return; }
        const int W = images_[0]->cols, H = images_[0]->rows;
        std::vector<float> img((size_t)num_frames_ * W * H * 3), P(num_frames_ * 16);
        for (size_t f = 0; f < num_frames_; ++f) {
            std::copy(images_[f]->data.begin(), images_[f]->data.end(), img.begin() + f * (size_t)W * H * 3);
            std::copy(poses_[f].begin(), poses_[f].end(), P.begin() + f * 16);
        }
        rc = psgsdf_set_keyframes(ctx_, (int)num_frames_, frame_idx_.data(), img.data(), W, H, P.data());
        if (rc) { fail("psgsdf_set_keyframes", rc); return; }
        rc = psgsdf_init(ctx_);
        if (rc) fail("psgsdf_init", rc);
    }

    // alternatingOptimize (PsOptimizer.cpp:239-428 / LedOptimizer.cpp:279-478): true = converged
    virtual bool alternatingOptimize(bool light, bool albedo, bool distance, bool pose) {
        PSG_STAGE("alternatingOptimize: total (incl. its dumps)");
        if (!ctx_) return false;
        if (lead_rank()) doc_.open((save_path_ + "optimizer_doc.txt").c_str());
        std::cout << "albation study settings: \n" << "light: " << light << "\n" << "albedo: " << albedo << "\n" << "distance: " << distance << "\n" << "pose: " << pose << std::endl;
        if (doc_.is_open()) doc_ << "albation study settings: \t" << "light: " << light << "\t" << "albedo: " << albedo << "\t" << "distance: " << distance << "\t" << "pose: " << pose
             << "\n" << "num of key frame: " << num_frames_ << " \n total voxels: " << num_voxels_ << "\n";
        int flags = (albedo ? PSGSDF_ALBEDO : 0) | (light ? PSGSDF_LIGHT : 0) | (distance ? PSGSDF_DIST : 0) | (pose ? PSGSDF_POSE : 0);
        std::vector<psgsdf_iter_stats> recs(settings_->max_it + 1);
        int n_done = 0, result = 0;
        psgsdf_set_record_observer(ctx_, &Optimizer::narrate_trampoline, this);
        psgsdf_set_on_iter_period(ctx_, 3);
        int rc = psgsdf_optimize(ctx_, flags, recs.data(), (int)recs.size(), &n_done, &result, &Optimizer::on_iter_trampoline, this);
        psgsdf_set_record_observer(ctx_, nullptr, nullptr);
        if (rc) return fail("psgsdf_optimize", rc);
        if (n_done > 0 && (recs[n_done - 1].converged || recs[n_done - 1].diverged)) {
            const int iter = n_done - 1;
            narrate(n_done, &recs[n_done - 1]);   // (the observer is not invoked for the terminating iteration.  Narration only: the reference returns at PsOptimizer.cpp:368-384, before ++iter and the iter % 3 dumps of :419-423 -- no after_iter_<k> files for it)
            std::cout << "===> [" << iter << "]: " << (result ? "converged!" : "diverged!") << std::endl;
            if (doc_.is_open()) doc_ << "===> [" << iter << "]: " << (result ? "converged! \n" : "diverged!\n");
            save_pointcloud("final_refined");                      // PsOptimizer.cpp:372-373,379-380
            extract_mesh("final_refined");
            if (settings_->model == LED) saveSDF("refined_sdf.sdf");   // LedOptimizer.cpp:422,430
        }
        // the reference mutates the shared settings (B9): report the effective weights back the same way
        psgsdf_info info{}; psgsdf_get_info(ctx_, &info);
        settings_->reg_weight_n = info.reg_weight_n; settings_->reg_weight_l = info.reg_weight_l;
        sync_back();
        if (doc_.is_open()) doc_.close();
        DumpQueue::get().drain();      // the reference has written its files when alternatingOptimize returns
        if (multi_rank()) { double one = 1; psgsdf_comm_allreduce_host(ctx_, &one, 1); }      // ... on every rank
        return result != 0;
    }

    // copy the refined state back into tSDF_ / poses_ (the reference mutates them in place)
    bool sync_back() {
        PSG_STAGE("sync back: refined volume to host");
        psgsdf_info info{}; psgsdf_get_info(ctx_, &info);
        size_t n = (size_t)info.dim[0] * info.dim[1] * info.dim[2];
        for (int a = 0; a < 3; ++a) tSDF_->grid_dim_[a] = info.dim[a];
        tSDF_->voxel_size_ = info.voxel_size; voxel_size_ = info.voxel_size;
        const bool no_volume = multi_rank() || skip_sync_back();      // (a rank holds a slab: no whole-volume host copy; nothing in voxelPS reads one after the optimisation)
        if (no_volume) n = 0;
        tSDF_->dist.resize(n); tSDF_->grad.resize(3 * n); tSDF_->weight.resize(n); tSDF_->rgb.resize(3 * n);
        tSDF_->vis.resize(n * info.vis_words); tSDF_->vis_words = info.vis_words;
        int rc = no_volume ? 0 : psgsdf_download_volume(ctx_, tSDF_->dist.data(), tSDF_->grad.data(), tSDF_->weight.data(), tSDF_->rgb.data(), tSDF_->vis.data());
        if (rc) return fail("psgsdf_download_volume", rc);
        std::vector<float> P(num_frames_ * 16);
        rc = psgsdf_download_poses(ctx_, P.data());
        if (rc) return fail("psgsdf_download_poses", rc);
        for (size_t f = 0; f < num_frames_; ++f) std::copy(P.begin() + 16 * f, P.begin() + 16 * f + 16, poses_[f].begin());
        return true;
    }

    // savePoses, OptimizerAux.cpp:580-599: "stamp tx ty tz qx qy qz qw" with Eigen's matrix->quaternion conversion
    bool savePoses(std::string filename) {
        PSG_STAGE("dump: poses");
        std::vector<float> P(num_frames_ * 16);
        if (psgsdf_download_poses(ctx_, P.data())) return false;
        if (!lead_rank()) return true;
        std::ofstream posefile((save_path_ + filename + ".txt").c_str());
        if (!posefile.is_open()) { std::cout << "couldn't save optimized poses! " << std::endl; return false; }
        for (size_t i = 0; i < num_frames_; ++i) {
            const float* M = &P[16 * i];
            float q[4];   // x y z w, Eigen::QuaternionBase::operator=(Matrix3)
            float t = M[0] + M[5] + M[10];
            if (t > 0) { t = std::sqrt(t + 1.0f); q[3] = 0.5f * t; t = 0.5f / t; q[0] = (M[9] - M[6]) * t; q[1] = (M[2] - M[8]) * t; q[2] = (M[4] - M[1]) * t; }
            else {
                int a = 0; if (M[5] > M[0]) a = 1; if (M[10] > M[a * 5]) a = 2;
                int b = (a + 1) % 3, c = (b + 1) % 3;
                t = std::sqrt(M[a * 5] - M[b * 5] - M[c * 5] + 1.0f);
                q[a] = 0.5f * t; t = 0.5f / t;
                q[3] = (M[c * 4 + b] - M[b * 4 + c]) * t; q[b] = (M[b * 4 + a] + M[a * 4 + b]) * t; q[c] = (M[c * 4 + a] + M[a * 4 + c]) * t;
            }
            posefile << (i < key_stamps_.size() ? key_stamps_[i] : std::to_string(i)) << " " << M[3] << " " << M[7] << " " << M[11] << " "
                     << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
        }
        return true;
    }

    // save_pointcloud, OptimizerAux.cpp:456-511 (grid-local coordinates: vox2float, origin not added)
    bool save_pointcloud(std::string filename) {
        PSG_STAGE("dump: point cloud");
        if (!host_writers()) return VolumetricGradSdf::device_pointcloud(ctx_, 0, save_path_ + filename + "_pointcloud.ply");
        psgsdf_info info{}; psgsdf_get_info(ctx_, &info);
        size_t n = (size_t)info.dim[0] * info.dim[1] * info.dim[2];
        std::vector<float> d(n), g(3 * n), rgb(3 * n); std::vector<int32_t> band(info.n_band);
        if (psgsdf_download_volume(ctx_, d.data(), g.data(), nullptr, rgb.data(), nullptr) || psgsdf_download_band(ctx_, band.data())) return false;
        std::ofstream ply((save_path_ + filename + "_pointcloud.ply").c_str());
        if (!ply.is_open()) { std::cout << " can't save point cloud!" << std::endl; return false; }
        size_t cnt = 0;
        for (int lin : band) if (std::abs(d[lin]) < std::sqrt(3.0) * info.voxel_size) ++cnt;
        ply << "ply\nformat ascii 1.0\nelement vertex " << cnt << "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
            << "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header" << std::endl;
        const int nx = info.dim[0], nxy = info.dim[0] * info.dim[1];
        for (int lin : band) {
            if (!(std::abs(d[lin]) < std::sqrt(3.0) * info.voxel_size)) continue;
            int k = lin / nxy, rest = lin - k * nxy, j = rest / nx, i = rest - j * nx;
            float gv[3] = {g[lin], g[n + lin], g[2 * n + lin]};
            float z = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2];
            if (z > 0) { float s = std::sqrt(z); gv[0] /= s; gv[1] /= s; gv[2] /= s; }
            float p[3] = {info.voxel_size * i - d[lin] * gv[0], info.voxel_size * j - d[lin] * gv[1], info.voxel_size * k - d[lin] * gv[2]};
            ply << p[0] << " " << p[1] << " " << p[2] << " " << gv[0] << " " << gv[1] << " " << gv[2] << " "
                << int(255 * rgb[lin]) << " " << int(255 * rgb[n + lin]) << " " << int(255 * rgb[2 * n + lin]) << std::endl;
        }
        return true;
    }

    // extract_mesh / saveSDF, OptimizerAux.cpp:278-363,513-577
    bool extract_mesh(std::string filename) {
        PSG_STAGE("dump: mesh (marching cubes + PLY)");
        if (!host_writers()) {
            const bool ok = VolumetricGradSdf::device_mesh(ctx_, save_path_ + filename + "_mesh.ply");
            if (!ok) std::cout << "couldn't save mesh " << save_path_ << filename << std::endl;
            return ok;
        }
        psgsdf_info info{}; psgsdf_get_info(ctx_, &info);
        size_t n = (size_t)info.dim[0] * info.dim[1] * info.dim[2];
        std::vector<float> d(n), w(n), rgb(3 * n);
        if (psgsdf_download_volume(ctx_, d.data(), nullptr, w.data(), rgb.data(), nullptr)) return false;
        bool ok = write_mesh(save_path_ + filename + "_mesh.ply", info.dim, info.voxel_size, d, w, rgb);
        if (!ok) std::cout << "couldn't save mesh " << save_path_ << filename << std::endl;
        return ok;
    }
    bool saveSDF(std::string filename) {
        PSG_STAGE("dump: sdf");
        if (!host_writers()) { psgsdf_info i2{}; psgsdf_get_info(ctx_, &i2); return VolumetricGradSdf::device_sdf(ctx_, save_path_ + filename, i2.voxel_size); }
        psgsdf_info info{}; psgsdf_get_info(ctx_, &info);
        size_t n = (size_t)info.dim[0] * info.dim[1] * info.dim[2];
        std::vector<float> d(n);
        if (psgsdf_download_volume(ctx_, d.data(), nullptr, nullptr, nullptr, nullptr)) return false;
        return write_sdf(save_path_ + filename, info.dim, info.voxel_size, d);
    }
    psgsdf_ctx* context() { return ctx_; }
};

// PsOptimizer.h / LedOptimizer.h: the constructor runs init() like the reference (PsOptimizer.cpp:15-23)
class PsOptimizer : public Optimizer {
public:
    PsOptimizer(VolumetricGradSdf* tSDF, const float voxel_size, const Mat3f& K, std::string save_path, OptimizerSettings* settings)
        : Optimizer(tSDF, voxel_size, K, save_path, settings) { init(); }
};
class LedOptimizer : public Optimizer {
public:
    LedOptimizer(VolumetricGradSdf* tSDF, const float voxel_size, const Mat3f& K, std::string save_path, OptimizerSettings* settings)
        : Optimizer(tSDF, voxel_size, K, save_path, settings) { settings_->model = LED; init(); }
};

}  // namespace psgsdf_host
