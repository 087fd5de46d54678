// voxelPS — drop-in for the reference executable: `voxelPS --config_file <config.json>` (cpp/voxel_ps/src/main_ps.cpp:41-343),
// same JSON keys (ConfigLoader.h:16-170), same exit codes, same output files under the "output" prefix.  Every numerical
// stage runs on the GPU through libpsgsdf.so: FALS normals, depth tracking, fusion, the photometric-stereo optimisation.
// Extra optional keys: "grid dim" (default 128, main_ps.cpp:123), "max keyframes" (default 40, :312).
#include "image_loader.hpp"
#include "mini_json.hpp"
#include <signal.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/wait.h>

using namespace psgsdf_host;

// compute_centroid, main_ps.cpp:346-375.  A float running sum over ~1e6 pixels, in the reference's order of operations (R * p, + t, += into the
// float accumulator; / float(counter)): the grid origin must be the reference's to the bit, not a better-rounded one.
static void compute_centroid(const Mat3f& K, const DepthImage& depth, const Mat4f& T, float out[3]) {
    float c[3] = {0.f, 0.f, 0.f}; int counter = 0;
    const float fx_inv = 1.f / K.v[0], fy_inv = 1.f / K.v[4], cx = K.v[2], cy = K.v[5];
    for (int y = 0; y < depth.rows; ++y) for (int x = 0; x < depth.cols; ++x) {
        float z = depth.data[(size_t)y * depth.cols + x];
        if (z > 0.0) {
            const float x0 = (float(x) - cx) * fx_inv, y0 = (float(y) - cy) * fy_inv;
            const float p[3] = {x0 * z, y0 * z, z};
            for (int a = 0; a < 3; ++a) { const float rp = (T[a * 4] * p[0] + T[a * 4 + 1] * p[1]) + T[a * 4 + 2] * p[2]; c[a] += rp + T[a * 4 + 3]; }
            ++counter;
        }
    }
    for (int a = 0; a < 3; ++a) out[a] = c[a] / float(counter);
}
// sampleKeyFrame, main_ps.cpp:392-421
template <class A, class B, class C_, class D>
static void sampleKeyFrame(A& frames, B& stamps, C_& images, D& poses, int max_num) {
    if ((int)frames.size() < max_num) return;
    max_num -= 1;
    float step = static_cast<float>(frames.size()) / static_cast<float>(max_num), idx = 0;
    A f2; B s2; C_ i2; D p2;
    for (int count = 0; count < max_num; ++count) { int i = static_cast<int>(idx); f2.push_back(frames[i]); s2.push_back(stamps[i]); i2.push_back(images[i]); p2.push_back(poses[i]); idx += step; }
    f2.push_back(frames.back()); s2.push_back(stamps.back()); i2.push_back(images.back()); p2.push_back(poses.back());
    frames = f2; stamps = s2; images = i2; poses = p2;
}
static void quat_of(const Mat4f& M, float q[4]) {   // Eigen::Quaternionf(Matrix3f): x y z w
    float t = M[0] + M[5] + M[10];
    if (t > 0) { t = std::sqrt(t + 1.0f); q[3] = 0.5f * t; t = 0.5f / t; q[0] = (M[9] - M[6]) * t; q[1] = (M[2] - M[8]) * t; q[2] = (M[4] - M[1]) * t; }
    else { int a = 0; if (M[5] > M[0]) a = 1; if (M[10] > M[a * 5]) a = 2; int b = (a + 1) % 3, c = (b + 1) % 3;
        t = std::sqrt(M[a * 5] - M[b * 5] - M[c * 5] + 1.0f); q[a] = 0.5f * t; t = 0.5f / t;
        q[3] = (M[c * 4 + b] - M[b * 4 + c]) * t; q[b] = (M[b * 4 + a] + M[a * 4 + b]) * t; q[c] = (M[c * 4 + a] + M[a * 4 + c]) * t; }
}

// host-side self tests that need no GPU (tests/test_host_tools.py): PNG decoding and the generated marching-cubes table
static int selftest_png(const char* path) {
    PngImage p; if (!read_png(path, p)) { std::cout << "FAIL" << std::endl; return 1; }
    unsigned long long sum = 0; for (uint16_t v : p.px) sum += v;
    std::cout << p.width << " " << p.height << " " << p.channels << " " << p.bit_depth << " " << sum << std::endl; return 0;
}
static int selftest_mc() {
    int dim[3] = {24, 24, 24}; float size[3] = {24, 24, 24}, org[3] = {0, 0, 0};
    MarchingCubes mc(dim, size, org);
    int ntri = 0; for (int c = 0; c < 256; ++c) ntri += (int)mc.table(c).size() / 3;
    std::vector<float> t(24 * 24 * 24), w(t.size(), 1.f); std::vector<unsigned char> col(t.size() + 2, 128);
    for (int k = 0; k < 24; ++k) for (int j = 0; j < 24; ++j) for (int i = 0; i < 24; ++i) { float dx = i - 11.3f, dy = j - 11.7f, dz = k - 11.1f; t[(k * 24 + j) * 24 + i] = 7.2f - std::sqrt(dx * dx + dy * dy + dz * dz); }
    mc.computeIsoSurface(t.data(), w.data(), col.data(), col.data(), col.data());
    // closedness by the divergence theorem: the signed volume must not depend on the origin it is taken from, and the area vectors sum to 0
    double vol = 0, vol2 = 0, area = 0, asum[3] = {0, 0, 0}, maxr = 0, minr = 1e9; auto& v = mc.vertices();
    const double o2[3] = {40.5, -17.25, 3.125};
    for (size_t f = 0; f + 2 < v.size(); f += 3) {
        double a[3], b[3], c[3], a2[3], b2[3], c2[3];
        for (int q = 0; q < 3; ++q) { a[q] = v[f][q]; b[q] = v[f + 1][q]; c[q] = v[f + 2][q]; a2[q] = a[q] - o2[q]; b2[q] = b[q] - o2[q]; c2[q] = c[q] - o2[q]; }
        vol += (a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0])) / 6.0;
        vol2 += (a2[0] * (b2[1] * c2[2] - b2[2] * c2[1]) - a2[1] * (b2[0] * c2[2] - b2[2] * c2[0]) + a2[2] * (b2[0] * c2[1] - b2[1] * c2[0])) / 6.0;
        double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double n[3] = {u[1] * w2[2] - u[2] * w2[1], u[2] * w2[0] - u[0] * w2[2], u[0] * w2[1] - u[1] * w2[0]};
        area += 0.5 * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); for (int q = 0; q < 3; ++q) asum[q] += 0.5 * n[q];
        for (int q = 0; q < 3; ++q) { double r = std::sqrt(std::pow(v[f + q][0] - 11.3, 2) + std::pow(v[f + q][1] - 11.7, 2) + std::pow(v[f + q][2] - 11.1, 2)); maxr = std::max(maxr, r); minr = std::min(minr, r); } }
    std::cout.precision(12);
    std::cout << ntri << " " << mc.num_faces() << " " << vol << " " << minr << " " << maxr << " " << vol2 << " " << area << " " << std::sqrt(asum[0] * asum[0] + asum[1] * asum[1] + asum[2] * asum[2]) << std::endl; return 0;
}
// the 256-case triangle table in use, one line per case: `case n_triangles e0 e1 e2 ...` (edge ids of the reference's numbering);
// generated = true: the first-principles table of round 1 (cross-check)
static int selftest_mc_table(bool generated) {
    int dim[3] = {4, 4, 4}; float size[3] = {4, 4, 4}, org[3] = {0, 0, 0};
    MarchingCubes mc(dim, size, org);
    for (int c = 0; c < 256; ++c) { const std::vector<int>& t = generated ? mc.generated_table(c) : mc.table(c); std::cout << c << " " << t.size() / 3; for (int e : t) std::cout << " " << e; std::cout << "\n"; }
    return 0;
}
// a mesh of a small analytic volume, written the way the product writes it: `--selftest-mc-ply N out.ply` (an off-centre bumpy sphere in an N^3 grid
// with a weight-0 corner region and varying colours): tests/test_host_tools.py re-derives it face by face
static int selftest_mc_ply(int n, const char* path) {
    int dim[3] = {n, n, n}; float size[3] = {0.5f * n, 0.5f * n, 0.5f * n}, org[3] = {0.1f, -0.2f, 0.3f};
    MarchingCubes mc(dim, size, org);
    std::vector<float> t((size_t)n * n * n), w(t.size(), 1.f); std::vector<unsigned char> r(t.size() + 2), g(t.size() + 2), b(t.size() + 2);
    for (int k = 0; k < n; ++k) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const size_t l = ((size_t)k * n + j) * n + i;
        const float dx = i - 0.47f * n, dy = j - 0.52f * n, dz = k - 0.45f * n;
        t[l] = 0.31f * n + 0.6f * std::sin(0.9f * i) * std::cos(0.7f * j + 0.3f * k) - std::sqrt(dx * dx + dy * dy + dz * dz);
        if (i + j + k < n / 2) w[l] = 0.f;
        r[l] = (unsigned char)((37 * i + 11 * j) & 255); g[l] = (unsigned char)((5 * j + 91 * k) & 255); b[l] = (unsigned char)((17 * k + 3 * i) & 255);
    }
    mc.computeIsoSurface(t.data(), w.data(), r.data(), g.data(), b.data());
    return mc.savePly(path) ? 0 : 1;
}

// the focus measure of one colour PNG as the keyframe selector computes it (SharpDetector.h:22-37), and the keyframe sub-sampling of
// main_ps.cpp:392-421 on the index list 0..n-1: tests/test_host_tools.py compares both with numpy / scipy restatements (no GPU needed)
static int selftest_lapm(const char* path) {
    ImageLoader* l = new MultiviewLoader("");
    ImageRGB c; if (!l->load_color(path, c)) { std::cout << "FAIL" << std::endl; return 1; }
    std::cout.precision(9); std::cout << modifiedLaplacian(c) << std::endl; delete l; return 0;
}
static int selftest_sample(int n, int max_num) {
    std::vector<int> frames(n), poses(n), images(n); std::vector<std::string> stamps(n);
    for (int i = 0; i < n; ++i) { frames[i] = 3 * i + 1; poses[i] = i; images[i] = -i; stamps[i] = std::to_string(i); }
    if ((int)frames.size() > max_num) sampleKeyFrame(frames, stamps, images, poses, max_num);      // (the call site's guard, main_ps.cpp:312)
    for (size_t i = 0; i < frames.size(); ++i) std::cout << frames[i] << ":" << stamps[i] << ":" << images[i] << ":" << poses[i] << (i + 1 < frames.size() ? " " : "\n");
    return 0;
}

// the threaded writers print floats with snprintf("%g"), the reference with `ostream << float`: the same characters for EVERY float?  `--selftest-fmt N` formats N
// pseudo-random bit patterns (all exponents, denormals, +-0, inf, nan) and a ladder of round-number cases both ways and counts the differences
static int selftest_fmt(long n) {
    unsigned long long x = 0x9e3779b97f4a7c15ull; long bad = 0, done = 0;
    auto check = [&](float v) {
        TextOut t; t.f(v);
        std::ostringstream o; o << v;
        if (o.str() != t.s) { if (bad < 5) std::cout << "MISMATCH " << o.str() << " vs " << t.s << std::endl; ++bad; }
        ++done;
    };
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (long i = 0; i < n; ++i) { uint32_t u = (uint32_t)(rnd() >> 16); float v; memcpy(&v, &u, 4); check(v); }
    // the ranges the dumps live in (the fast path of format_g6): uniform in (-2, 2), log-uniform over 1e-7 .. 1e7, and decimals of seven digits ending in 5
    // -- the floats nearest to a rounding tie of the sixth digit
    for (long i = 0; i < n; ++i) check((float)((double)(rnd() >> 11) * (4.0 / 9007199254740992.0) - 2.0));
    for (long i = 0; i < n; ++i) { const double u = (double)(rnd() >> 11) / 9007199254740992.0; check((float)std::pow(10.0, 14.0 * u - 7.0)); }
    for (long i = 0; i < n; ++i) { const long k = (long)(rnd() % 900000) + 100000; const int e = (int)(rnd() % 16) - 9; const double t = ((double)k * 10.0 + 5.0) * std::pow(10.0, (double)e - 6.0); check((float)t); check(std::nextafter((float)t, 0.0f)); check(std::nextafter((float)t, 1e30f)); }
    for (int e = -45; e <= 38; ++e) for (float m : {1.0f, 9.999995f, 9.9999995f, 1.234565f, 1.2345649f, 0.5f, 2.5f}) { check(m * std::pow(10.0f, (float)e)); check(-m * std::pow(10.0f, (float)e)); }
    for (int i = 0; i <= 255; ++i) check((float)i * (1.0f / 255.0f));
    std::cout << done << " " << bad << std::endl;
    return bad ? 1 : 0;
}

// `voxelPS --config_file <cfg> --gpus N [--transport rccl|sockets]`: this process only starts the N ranks (itself, once per GPU) and waits for them.
// rccl (default): rank r runs on device r, the ranks meet in ncclCommInitRank over the id made here.  sockets: the engine's node-local transport over one
// socket pair per pair of ranks (psgsdf_comm_init_sockets) -- a node without a working RCCL, or ranks that share a device: VOXELPS_SHARE_GPU=1 puts
// all of them on device 0, VOXELPS_CU_MASKS="0:128,128:256" confines rank r to the r-th CU range (the one-GPU rehearsal, tests/test_voxelps_ranks_gpu.py).
// A rank that fails ends the run: the others are terminated (exactly the processes started here) and its exit code is returned.
static int launch_ranks(int n, bool sockets, int argc, char* argv[]) {
    if (n < 2 || n > 32) { std::cerr << "--gpus " << n << ": 2 .. 32 ranks" << std::endl; return 1; }
    std::vector<std::vector<int>> mesh(n, std::vector<int>(n, -1));
    std::string id_hex;
    if (sockets) {
        for (int r = 0; r < n; ++r) for (int q = r + 1; q < n; ++q) { int sv[2]; if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; } mesh[r][q] = sv[0]; mesh[q][r] = sv[1]; }
    } else {
        uint8_t id[128];
        if (psgsdf_comm_unique_id(id)) { std::cerr << "no RCCL on this node (psgsdf_comm_unique_id failed): --transport sockets runs without it" << std::endl; return 1; }
        static const char* hx = "0123456789abcdef";
        for (int i = 0; i < 128; ++i) { id_hex.push_back(hx[id[i] >> 4]); id_hex.push_back(hx[id[i] & 15]); }
    }
    std::vector<std::string> masks;
    if (const char* e = getenv("VOXELPS_CU_MASKS")) { std::string m = e; size_t a = 0; while (a <= m.size()) { size_t b = m.find(',', a); if (b == std::string::npos) b = m.size(); masks.push_back(m.substr(a, b - a)); a = b + 1; } }
    std::vector<pid_t> pids;
    for (int r = 0; r < n; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); for (pid_t p : pids) kill(p, SIGTERM); return 1; }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGTERM);      // a launcher that is killed takes its ranks with it
            for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (a != r && mesh[a][b] >= 0) close(mesh[a][b]);      // the other ranks' ends
            if ((int)masks.size() == n) setenv("PSGSDF_CU_MASK", masks[r].c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
            std::vector<std::string> av(argv, argv + argc);
            av.push_back("--rank"); av.push_back(std::to_string(r)); av.push_back("--nranks"); av.push_back(std::to_string(n));
            if (sockets) { std::string f; for (int q = 0; q < n; ++q) f += (q ? "," : "") + std::to_string(mesh[r][q]); av.push_back("--fds"); av.push_back(f); }
            else { av.push_back("--comm-id"); av.push_back(id_hex); }
            std::vector<char*> cv; for (auto& a : av) cv.push_back(const_cast<char*>(a.c_str())); cv.push_back(nullptr);
            execv("/proc/self/exe", cv.data());
            perror("execv"); _exit(127);
        }
        pids.push_back(pid);
    }
    for (auto& row : mesh) for (int f : row) if (f >= 0) close(f);
    int rc = 0, left = n;
    while (left > 0) {
        int st = 0; const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) break;
        auto it = std::find(pids.begin(), pids.end(), p); if (it == pids.end()) continue;
        --left; *it = -1;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code != 0 && rc == 0) {
            rc = code;
            std::cerr << "rank " << (it - pids.begin()) << " ended with " << code << ": stopping the other ranks" << std::endl;
            for (pid_t q : pids) if (q > 0) kill(q, SIGTERM);
        }
    }
    return rc;
}

int main(int argc, char* argv[]) {
    if (argc >= 3 && std::string(argv[1]) == "--selftest-fmt") return selftest_fmt(atol(argv[2]));
    if (argc >= 3 && std::string(argv[1]) == "--selftest-lapm") return selftest_lapm(argv[2]);
    if (argc >= 4 && std::string(argv[1]) == "--selftest-sample") return selftest_sample(atoi(argv[2]), atoi(argv[3]));
    if (argc >= 3 && std::string(argv[1]) == "--selftest-png") return selftest_png(argv[2]);
    if (argc >= 2 && std::string(argv[1]) == "--selftest-mc") return selftest_mc();
    if (argc >= 2 && std::string(argv[1]) == "--selftest-mc-table") return selftest_mc_table(false);
    if (argc >= 2 && std::string(argv[1]) == "--selftest-mc-generated") return selftest_mc_table(true);
    if (argc >= 4 && std::string(argv[1]) == "--selftest-mc-ply") return selftest_mc_ply(atoi(argv[2]), argv[3]);
    int want_ranks = 1; std::string transport = "rccl";
    std::string configfile, timing_file;      // --timing <file.json>: wall-clock per stage (no reference counterpart; the reference's outputs are unchanged)
    for (int i = 1; i < argc; ++i) { std::string a = argv[i]; if (a == "--config_file" && i + 1 < argc) configfile = argv[++i]; else if (a.rfind("--config_file=", 0) == 0) configfile = a.substr(14);
        else if (a == "--timing" && i + 1 < argc) timing_file = argv[++i];
        else if (a == "--host-writers") host_writers() = true;
        else if (a == "--frame-solver" && i + 1 < argc) { setenv("PSGSDF_FRAME_SOLVE", argv[++i], 1); std::cout << "frame solver: " << argv[i] << std::endl; }      // eigen = the reference's own solver of the light / pose blocks (include/psgsdf.h psgsdf_set_frame_solver); default ldlt.  The ranks of --gpus N inherit the environment.
        else if (a == "--gpus" && i + 1 < argc) want_ranks = atoi(argv[++i]);
        else if (a == "--transport" && i + 1 < argc) transport = argv[++i];
        else if (a == "--rank" && i + 1 < argc) rank_info().rank = atoi(argv[++i]);
        else if (a == "--nranks" && i + 1 < argc) rank_info().n = atoi(argv[++i]);
        else if (a == "--fds" && i + 1 < argc) { std::string f = argv[++i]; size_t p0 = 0; while (p0 <= f.size()) { size_t p1 = f.find(',', p0); if (p1 == std::string::npos) p1 = f.size(); rank_info().fds.push_back(atoi(f.substr(p0, p1 - p0).c_str())); p0 = p1 + 1; } }
        else if (a == "--comm-id" && i + 1 < argc) { std::string h = argv[++i]; for (size_t q = 0; q + 1 < h.size(); q += 2) rank_info().id.push_back((uint8_t)strtol(h.substr(q, 2).c_str(), nullptr, 16)); } }      // round 4's dump path (dense download, host marching cubes, iostream) and frame path (serial decode, normals through the host): the cross-check of the device / threaded one
    if (transport != "rccl" && transport != "sockets") { std::cerr << "--transport: rccl or sockets" << std::endl; return 1; }
    if ((want_ranks > 1 || multi_rank()) && host_writers()) { std::cerr << "--host-writers is the single-process cross-check" << std::endl; return 1; }
    if (want_ranks > 1 && !multi_rank()) return launch_ranks(want_ranks, transport == "sockets", argc, argv);
    static std::ofstream null_out;
    if (multi_rank()) {
        RankInfo& ri = rank_info();
        ri.sockets = transport == "sockets";
        ri.device = getenv("VOXELPS_SHARE_GPU") && atoi(getenv("VOXELPS_SHARE_GPU")) ? 0 : ri.rank;
        if (ri.rank < 0 || ri.rank >= ri.n || (ri.sockets ? (int)ri.fds.size() != ri.n : ri.id.size() != 128)) { std::cerr << "--rank / --nranks / --fds / --comm-id are set by `voxelPS --gpus N`" << std::endl; return 1; }
        if (!lead_rank()) { null_out.open("/dev/null"); std::cout.rdbuf(null_out.rdbuf()); }      // rank 0 narrates
    }
    const auto t_main0 = std::chrono::steady_clock::now();
    std::cout << "load the config file from: " << configfile << std::endl;
    JsonObject config;
    if (!config.load(configfile)) { std::cout << "can't load config file!" << std::endl << "fail to load the config file!" << std::endl; return 1; }
    if (!config.contains("input") || !config.contains("output") || !config.contains("datatype")) {
        std::cout << "missing necessary input arguments (input/out folder/datatype) in config file!" << std::endl << "fail to load the config file!" << std::endl; return 1; }
    const std::string input = config.str("input"), output = config.str("output"), datatype = config.str("datatype");
    ImageLoader* loader;
    if (datatype == "tum") loader = new TumrgbdLoader(input);
    else if (datatype == "led" || datatype == "synth") loader = new SynthLoader(input);
    else if (datatype == "intrinsic3d" || datatype == "multiview") loader = new MultiviewLoader(input);
    else { std::cerr << "Your specified dataset type is not supported (yet)." << std::endl; return 1; }
    // TrackingSettings.h:26-38 defaults
    std::string pose_file = "pose.txt"; size_t first = 0, last = (size_t)-1; float voxel_size = 0.02f, truncation_factor = 5, zmin = 0.5f, zmax = 3.5f, sharp_thr = 0.5f;
    if (config.contains("pose filename")) pose_file = config.str("pose filename");
    if (config.contains("first")) first = (size_t)config.num("first");
    if (config.contains("last")) last = (size_t)config.num("last");
    if (config.contains("voxel size")) voxel_size = (float)config.num("voxel size");
    if (config.contains("truncation factor")) truncation_factor = (float)config.num("truncation factor");
    if (config.contains("sharpness threshold")) sharp_thr = (float)config.num("sharpness threshold");
    if (config.contains("zmin")) zmin = (float)config.num("zmin");
    if (config.contains("zmax")) zmax = (float)config.num("zmax");
    OptimizerSettings* opt_set_ = new OptimizerSettings();
    if (config.contains("model type")) {
        std::string m = config.str("model type");
        if (m == "SH1") { opt_set_->model = SH1; opt_set_->order = 1; } else if (m == "SH2") { opt_set_->model = SH2; opt_set_->order = 2; } else if (m == "LED") opt_set_->model = LED;
        else { std::cerr << "Your specified model type is not supported (yet)." << std::endl; return 1; }
    }
    if (config.contains("loss function")) {
        std::string l = config.str("loss function");
        if (l == "cauchy") opt_set_->loss = CAUCHY; else if (l == "l2") opt_set_->loss = L2; else if (l == "huber") opt_set_->loss = HUBER;
        else if (l == "trunc_l2") opt_set_->loss = TRUNC_L2;   // the reference leaves the loss uninitialised here (ConfigLoader.h:126, B7); we honour the key
        else if (l == "tukey") opt_set_->loss = TUKEY;
        else { std::cerr << "Your specified loss function type is not supported (yet)." << std::endl; return 1; }
    }
    if (config.contains("reg albedo")) opt_set_->reg_weight_rho = (float)config.num("reg albedo");
    if (config.contains("reg norm")) opt_set_->reg_weight_n = (float)config.num("reg norm");
    if (config.contains("reg laplacian")) opt_set_->reg_weight_l = (float)config.num("reg laplacian");
    if (config.contains("max iter")) opt_set_->max_it = (int)config.num("max iter");
    if (config.contains("damping")) opt_set_->damping = (float)config.num("damping");
    if (config.contains("converge threshold")) opt_set_->conv_threshold = (float)config.num("converge threshold");
    if (config.contains("upsample")) opt_set_->upsample = config.boolean("upsample");
    if (config.contains("lambda")) { opt_set_->lambda = (float)config.num("lambda"); opt_set_->lambda_sq = opt_set_->lambda * opt_set_->lambda; }
    if (lead_rank()) { std::ofstream save_conf(output + "saved_config.json"); if (!save_conf.is_open()) std::cout << "could not save config file." << std::endl; config.dump(save_conf); }
    bool light = false, albedo = false, distance = false, pose = false;
    if (config.contains("--light")) light = config.boolean("--light");
    if (config.contains("--albedo")) albedo = config.boolean("--albedo");
    if (config.contains("--distance")) distance = config.boolean("--distance");
    if (config.contains("--pose")) pose = config.boolean("--pose");
    const int grid = config.contains("grid dim") ? (int)config.num("grid dim") : 128;
    const int max_key = config.contains("max keyframes") ? (int)config.num("max keyframes") : 40;
    const float truncation = truncation_factor * voxel_size;

    if (!loader->load_intrinsics("intrinsics.txt")) { std::cerr << "No intrinsics file found in " << input << "!" << std::endl; return 1; }
    const Mat3f K = loader->K();
    ImageRGB color; DepthImage depth;
    if (!loader->load_next(color, depth)) { std::cerr << " -> Frame could not be loaded!" << std::endl; return 1; }
    if (color.rows != depth.rows || color.cols != depth.cols) { std::cerr << "-> depth image and color image sizes don't match." << std::endl; return 1; }
    loader->reset_counter();

    int grid_dim[3] = {grid, grid, grid};
    VolumetricGradSdf* tSDF = new VolumetricGradSdf();
    RigidPointOptimizer* pOpt = nullptr; Optimizer* vOpt = nullptr;
    std::ofstream pose_out; if (lead_rank()) pose_out.open(output + "tracking_poses.txt");
    std::vector<Mat4f> poses; std::vector<int> keyframes{0}; std::vector<std::string> key_stamps; std::vector<std::shared_ptr<ImageRGB>> key_images;
    const Mat4f I4 = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<Mat4f> key_poses{I4};   // key_poses[0] is Identity even with GT poses (main_ps.cpp:139, B1)
    int dist_to_last_keyframe = 0; bool GT_pose = false;
    if (!loader->load_pose(pose_file, poses)) { std::cout << "GT poses is not avalible!" << std::endl; poses.push_back(I4); }
    else { std::cout << poses.size() << " GT poses are loaded." << std::endl; GT_pose = true; }
    { std::string d, r; for (size_t i = 0; i < first; ++i) if (host_writers()) loader->load_next(color, depth); else loader->next_names(d, r); }      // (frames in front of "first" are skipped; the reference decodes them, nothing reads them)
    Mat4f cur_pose = I4;
    size_t window = host_writers() ? 1 : std::min<size_t>(4, std::max(2u, std::thread::hardware_concurrency() / 2));
    if (const char* e = getenv("VOXELPS_PREFETCH")) window = (size_t)std::max(1, atoi(e));      // (frames decoded ahead of the fusion)
    FramePrefetcher prefetch(loader, window);
    std::string stamp_rgb, stamp_depth;
    for (size_t i = first; i <= last; ++i) {
        std::cout << "Working on frame: " << i << std::endl;
        {
            PSG_STAGE("decode: PNG colour + depth (host; wait for the prefetched frame)");
            if (host_writers()) { if (!loader->load_next(color, depth)) { std::cerr << " -> Frame " << i << " could not be loaded!" << std::endl; break; } stamp_rgb = loader->rgb_timestamp(); stamp_depth = loader->depth_timestamp(); }
            else {
                auto fr = prefetch.next();
                if (!fr) { std::cerr << " -> Frame " << i << " could not be loaded!" << std::endl; break; }
                color = std::move(fr->color); depth = std::move(fr->depth); stamp_rgb = fr->rgb_stamp; stamp_depth = fr->depth_stamp;
            }
        }
        if (GT_pose && i >= poses.size()) break;
        if (i == first) {
            float centroid[3]; compute_centroid(K, depth, poses[0], centroid);
            psgsdf_settings s{};
            s.model = opt_set_->model == LED ? PSGSDF_LED : (opt_set_->model == SH2 ? PSGSDF_SH2 : PSGSDF_SH1); s.loss = (int)opt_set_->loss; s.lambda = opt_set_->lambda; s.damping = opt_set_->damping;
            s.reg_weight_rho = opt_set_->reg_weight_rho; s.reg_weight_n = opt_set_->reg_weight_n; s.reg_weight_l = opt_set_->reg_weight_l; s.max_it = opt_set_->max_it;
            s.conv_threshold = opt_set_->conv_threshold; s.upsample = opt_set_->upsample ? 1 : 0; s.ref_quirks = 1;
            // frames to expect: the config's range when it has a "last", else unknown (the reference runs until the loader is exhausted,
            // main_ps.cpp:222-258) -- the device volume widens its per-voxel visibility words when the sequence outgrows this
            const int max_frames = last == (size_t)-1 ? 256 : (int)std::min<size_t>(last - first + 1, (size_t)1 << 20);
            if (!tSDF->attach(grid_dim, voxel_size, centroid, truncation, K, s, max_frames)) { std::cerr << "could not create the device volume" << std::endl; return 1; }
            tSDF->set_zmin(zmin); tSDF->set_zmax(zmax);
            pOpt = new RigidPointOptimizer(tSDF);
            if (opt_set_->model == LED) vOpt = new LedOptimizer(tSDF, voxel_size, K, output, opt_set_); else vOpt = new PsOptimizer(tSDF, voxel_size, K, output, opt_set_);
            if (!tSDF->update(color, depth.data, poses[0])) { std::cerr << " -> Frame " << i << " could not be fused: " << tSDF->last_error() << std::endl; return 1; }
            cur_pose = poses[0];
            key_stamps.push_back(stamp_rgb);
            key_images.push_back(std::make_shared<ImageRGB>(color));
        } else {
            tSDF->increase_counter();
            bool integrated = false;
            if (GT_pose) { cur_pose = poses[i]; integrated = tSDF->update(color, depth.data, poses[i]); }
            else { bool conv = pOpt->optimize(depth.data, depth.cols, depth.rows); cur_pose = pOpt->pose(); if (conv) integrated = tSDF->update(color, depth.data, cur_pose); }
            if (!integrated && tSDF->last_error()[0]) { std::cerr << " -> Frame " << i << " could not be fused: " << tSDF->last_error() << std::endl; return 1; }   // a frame that is not in the volume must not become a keyframe
            if (integrated) {
                bool sharp; { PSG_STAGE("keyframe selection: focus measure (host)"); sharp = sharpDetector(color, sharp_thr); }
                if (sharp || dist_to_last_keyframe > 5) {
                    dist_to_last_keyframe = 0;
                    keyframes.push_back((int)(i - first)); key_stamps.push_back(stamp_rgb); key_poses.push_back(cur_pose);
                    key_images.push_back(std::make_shared<ImageRGB>(color));
                } else ++dist_to_last_keyframe;
            }
        }
        float q[4]; quat_of(cur_pose, q);
        pose_out << stamp_depth << " " << cur_pose[3] << " " << cur_pose[7] << " " << cur_pose[11] << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
    }
    pose_out.close();
    if (!vOpt) { std::cerr << "no frame was processed" << std::endl; return 1; }
    if (!tSDF->extract_mesh(output + "init_mesh.ply")) std::cerr << "Could not save mesh to " << output + "init_mesh.ply" << "!" << std::endl;
    if (!tSDF->extract_pc(output + "init_pointcloud.ply")) std::cerr << "Could not save point cloud to " << output + "init_pointcloud.ply" << "!" << std::endl;
    if (!tSDF->saveSDF(output + "init_sdf.sdf")) std::cerr << "Could not save sdf to " << output + "init_sdf.sdf" << "!" << std::endl;
    std::cout << " selected key frame: " << std::endl; for (int k : keyframes) std::cout << k << " "; std::cout << std::endl;
    if ((int)keyframes.size() > max_key) sampleKeyFrame(keyframes, key_stamps, key_images, key_poses, max_key);
    std::cout << " selected key frame after sampling: " << std::endl; for (int k : keyframes) std::cout << k << " "; std::cout << std::endl;
    skip_sync_back() = !host_writers();
    vOpt->setImages(key_images); vOpt->setKeyframes(keyframes); vOpt->setKeytimestamps(key_stamps); vOpt->setPoses(key_poses);
    vOpt->init();
    vOpt->alternatingOptimize(light, albedo, distance, pose);
    DumpQueue::get().drain();
    const int write_failures = DumpQueue::get().failures();      // (a rank whose piece of a shared output file could not be written: the file has a hole where it belongs)
    if (write_failures) std::cerr << "rank " << rank_info().rank << ": " << write_failures << " output file piece(s) could not be written" << std::endl;
    delete vOpt; delete pOpt; delete tSDF; delete loader; delete opt_set_;
    if (!timing_file.empty() && lead_rank()) {
        char extra[256];
        snprintf(extra, sizeof(extra), ", \"total_s\": %.6f, \"frames\": %zu, \"keyframes\": %zu", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main0).count(),
                 (size_t)(last == (size_t)-1 ? 0 : last - first + 1), keyframes.size());
        StageClock::get().write_json(timing_file, extra);
    }
    return write_failures ? 1 : 0;
}
