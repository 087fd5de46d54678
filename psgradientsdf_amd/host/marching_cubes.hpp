// marching_cubes.hpp — iso-surface extraction + the reference's ASCII PLY mesh layout (third/mesh/MarchingCubes.cpp:
// computeIsoSurface :314-505, computeLutIndex :511-556, interpolate :559-579, getColor :592-608, computeTriangles :611-637,
// savePly :659-699).  Corner / edge numbering, the inside test (tsdf > iso), the per-edge interpolation, the colour
// look-up with its +1/+2 index offsets, the dim-2 loop bounds and the non-indexed vertices are the
// reference's, and so are the triangles: the 256 x 16 triangle table is the classic one (data: mc_tritable.inc, extracted by
// tests/golden/make_mc_tritable.py and checked against tests/golden/mc_tritable.npy), walked in its order, so a mesh written
// from the same volume has the reference's faces in the reference's order.  The first-principles
// generator of round 1 (face contours -> closed loops -> fans) is kept as a cross-check of the table (generated_table():
// same crossed edges, same orientation; tests/test_host_tools.py).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

namespace psgsdf_host {

class MarchingCubes {
    // corner c -> (dx,dy,dz) in the reference's numbering (computeLutIndex: bit0 = (i+1,j+1,k), bit1 = (i+1,j,k), ...)
    static constexpr int kCorner[8][3] = {{1, 1, 0}, {1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}, {0, 1, 1}};
    static constexpr int kEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    static constexpr signed char kTriTable[256][16] = {
#include "mc_tritable.inc"
    };
    std::array<std::vector<int>, 256> tri_;   // edge triples per case (kTriTable without the -1 padding)
    std::array<std::vector<int>, 256> gen_;   // the generated table (cross-check only)

    static int edge_between(int a, int b) { for (int e = 0; e < 12; ++e) if ((kEdge[e][0] == a && kEdge[e][1] == b) || (kEdge[e][0] == b && kEdge[e][1] == a)) return e; return -1; }
    void build_table() {
        // faces as corner cycles
        static const int faces[6][4] = {{0, 1, 2, 3}, {4, 7, 6, 5}, {0, 4, 5, 1}, {1, 5, 6, 2}, {2, 6, 7, 3}, {3, 7, 4, 0}};
        for (int cs = 1; cs < 255; ++cs) {
            std::vector<std::pair<int, int>> seg;   // pairs of crossed edges joined on some face
            for (auto& fc : faces) {
                int in[4]; int nin = 0; for (int q = 0; q < 4; ++q) { in[q] = (cs >> fc[q]) & 1; nin += in[q]; }
                int e[4]; for (int q = 0; q < 4; ++q) e[q] = edge_between(fc[q], fc[(q + 1) & 3]);   // edge q joins corner q and q+1
                std::vector<int> crossed; for (int q = 0; q < 4; ++q) if (in[q] != in[(q + 1) & 3]) crossed.push_back(q);
                if (crossed.size() == 2) seg.push_back({e[crossed[0]], e[crossed[1]]});
                else if (crossed.size() == 4) {   // ambiguous face: cut off each inside corner separately
                    for (int q = 0; q < 4; ++q) if (in[q]) seg.push_back({e[(q + 3) & 3], e[q]});
                }
            }
            std::vector<char> used(seg.size(), 0);
            for (size_t s0 = 0; s0 < seg.size(); ++s0) {
                if (used[s0]) continue;
                std::vector<int> loop{seg[s0].first, seg[s0].second}; used[s0] = 1;
                while (loop.back() != loop.front()) {
                    bool found = false;
                    for (size_t s = 0; s < seg.size() && !found; ++s) {
                        if (used[s]) continue;
                        if (seg[s].first == loop.back()) { loop.push_back(seg[s].second); used[s] = 1; found = true; }
                        else if (seg[s].second == loop.back()) { loop.push_back(seg[s].first); used[s] = 1; found = true; }
                    }
                    if (!found) break;
                }
                loop.pop_back();
                if (loop.size() < 3) continue;
                // orientation: normals point from the inside (bit set) to the outside
                auto mid = [&](int e, double* m) { for (int a = 0; a < 3; ++a) m[a] = 0.5 * (kCorner[kEdge[e][0]][a] + kCorner[kEdge[e][1]][a]); };
                double nrm[3] = {0, 0, 0};
                for (size_t q = 0; q < loop.size(); ++q) { double a[3], b[3]; mid(loop[q], a); mid(loop[(q + 1) % loop.size()], b);
                    nrm[0] += (a[1] - b[1]) * (a[2] + b[2]); nrm[1] += (a[2] - b[2]) * (a[0] + b[0]); nrm[2] += (a[0] - b[0]) * (a[1] + b[1]); }
                double out = 0;
                for (int e : loop) { int a = kEdge[e][0], b = kEdge[e][1]; if (!((cs >> a) & 1)) std::swap(a, b);   // a inside, b outside
                    for (int k = 0; k < 3; ++k) out += nrm[k] * (kCorner[b][k] - kCorner[a][k]); }
                if (out < 0) std::reverse(loop.begin(), loop.end());
                for (size_t q = 1; q + 1 < loop.size(); ++q) { gen_[cs].push_back(loop[0]); gen_[cs].push_back(loop[q]); gen_[cs].push_back(loop[q + 1]); }
            }
        }
    }

    int dim_[3]; float size_[3], voxel_[3], origin_[3];
    const float* tsdf_ = nullptr; const float* w_ = nullptr; const unsigned char *r_ = nullptr, *g_ = nullptr, *b_ = nullptr;
    size_t total_ = 0;
    std::vector<std::array<float, 3>> vertices_; std::vector<std::array<unsigned char, 3>> colors_; std::vector<std::array<int, 3>> faces_;

    size_t lin(int i, int j, int k) const { return (size_t)k * dim_[0] * dim_[1] + (size_t)j * dim_[0] + i; }
    // MarchingCubes.cpp:559-579
    static void interpolate(float t0, float t1, const float* v0, const float* v1, float iso, float* out) {
        if (std::fabs(iso - t0) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v0[a]; return; }
        if (std::fabs(iso - t1) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v1[a]; return; }
        if (std::fabs(t0 - t1) < 1e-7) { for (int a = 0; a < 3; ++a) out[a] = v0[a]; return; }
        double mu = (iso - t0) / (t1 - t0);
        if (mu > 1.0) mu = 1.0; else if (mu < 0) mu = 0.0;
        for (int a = 0; a < 3; ++a) out[a] = (float)(v0[a] + mu * (v1[a] - v0[a]));
    }
    unsigned char at(const unsigned char* p, size_t i) const { return i < total_ ? p[i] : 0; }   // the +1/+2 look-ups run one/two past the last voxel (B10)

public:
    // dimensions, physical size (voxel size = size / dim, MarchingCubes.h ctor) and origin offset (subtracted)
    MarchingCubes(const int dim[3], const float size[3], const float origin[3]) {
        for (int a = 0; a < 3; ++a) { dim_[a] = dim[a]; size_[a] = size[a]; origin_[a] = origin[a]; voxel_[a] = size[a] / dim[a]; }
        total_ = (size_t)dim[0] * dim[1] * dim[2];
        for (int cs = 0; cs < 256; ++cs) for (int i = 0; i < 16 && kTriTable[cs][i] >= 0; ++i) tri_[cs].push_back(kTriTable[cs][i]);
    }
    const std::vector<int>& table(int cs) const { return tri_[cs]; }
    const std::vector<int>& generated_table(int cs) { if (gen_[1].empty()) build_table(); return gen_[cs]; }
    // corner / edge geometry for the tests
    static const int (*corners())[3] { return kCorner; }
    static const int (*edges())[2] { return kEdge; }

    bool computeIsoSurface(const float* tsdf, const float* weights, const unsigned char* red, const unsigned char* green, const unsigned char* blue, float iso = 0.0f) {
        if (!tsdf || !weights || !red || !green || !blue) return false;
        tsdf_ = tsdf; w_ = weights; r_ = red; g_ = green; b_ = blue;
        vertices_.clear(); colors_.clear(); faces_.clear();
        for (int z = 0; z < dim_[2] - 2; ++z) for (int y = 0; y < dim_[1] - 2; ++y) for (int x = 0; x < dim_[0] - 2; ++x) {
            size_t off[8]; bool valid = true; int cs = 0;
            for (int c = 0; c < 8; ++c) { off[c] = lin(x + kCorner[c][0], y + kCorner[c][1], z + kCorner[c][2]); if (w_[off[c]] == 0.0f) valid = false; }
            if (!valid) continue;
            for (int c = 0; c < 8; ++c) if (tsdf_[off[c]] > iso) cs |= 1 << c;
            if (cs == 0 || cs == 255) continue;
            float ep[12][3]; unsigned char ec[12][3]; bool have[12] = {false};
            for (int e : tri_[cs]) {
                if (have[e]) continue;
                have[e] = true;
                const int a = kEdge[e][0], b = kEdge[e][1];
                float pa[3], pb[3];
                for (int k = 0; k < 3; ++k) { int ia = (k == 0 ? x : k == 1 ? y : z) + kCorner[a][k], ib = (k == 0 ? x : k == 1 ? y : z) + kCorner[b][k];
                    pa[k] = ia * voxel_[k] - origin_[k]; pb[k] = ib * voxel_[k] - origin_[k]; }   // voxelToWorld, :647-651
                interpolate(tsdf_[off[a]], tsdf_[off[b]], pa, pb, iso, ep[e]);
                // getColor :592-608; for edges 2, 3, 6 and 7 the reference passes the two end points to getColor in the OPPOSITE order to getVertex
                // (:371, :383, :419, :431), which matters for the rounding of the interpolated byte
                const bool rev = e == 2 || e == 3 || e == 6 || e == 7;
                const size_t o1 = rev ? off[b] : off[a], o2 = rev ? off[a] : off[b];
                float ca[3] = {at(r_, o1) / 255.0f, at(g_, o1 + 1) / 255.0f, at(b_, o1 + 2) / 255.0f};   // :598-599
                float cb[3] = {at(r_, o2) / 255.0f, at(g_, o2 + 1) / 255.0f, at(b_, o2 + 2) / 255.0f};
                float cv[3]; interpolate(tsdf_[o1], tsdf_[o2], ca, cb, iso, cv);
                for (int k = 0; k < 3; ++k) ec[e][k] = (unsigned char)(cv[k] * 255.0f);   // (cVal * 255.0).cast<unsigned char>(): a float product (Eigen casts the scalar), truncated
            }
            for (size_t i = 0; i + 2 < tri_[cs].size(); i += 3) {
                const int e0 = tri_[cs][i], e1 = tri_[cs][i + 1], e2 = tri_[cs][i + 2];
                auto same = [&](int p, int q) { return ep[p][0] == ep[q][0] && ep[p][1] == ep[q][1] && ep[p][2] == ep[q][2]; };
                if (same(e0, e1) || same(e0, e2) || same(e1, e2)) continue;   // degenerate, :623
                int base = (int)vertices_.size();
                for (int e : {e0, e1, e2}) { vertices_.push_back({ep[e][0], ep[e][1], ep[e][2]}); colors_.push_back({ec[e][0], ec[e][1], ec[e][2]}); }
                faces_.push_back({base, base + 1, base + 2});
            }
        }
        return true;
    }
    size_t num_faces() const { return faces_.size(); }
    const std::vector<std::array<float, 3>>& vertices() const { return vertices_; }
    // MarchingCubes.cpp:659-699
    bool savePly(const std::string& filename) const {
        if (vertices_.empty()) return false;
        std::ofstream ply(filename.c_str());
        if (!ply.is_open()) return false;
        ply << "ply" << std::endl << "format ascii 1.0" << std::endl << "element vertex " << vertices_.size() << std::endl
            << "property float x" << std::endl << "property float y" << std::endl << "property float z" << std::endl
            << "property uchar red" << std::endl << "property uchar green" << std::endl << "property uchar blue" << std::endl
            << "element face " << (int)faces_.size() << std::endl << "property list uchar int vertex_indices" << std::endl << "end_header" << std::endl;
        for (size_t i = 0; i < vertices_.size(); ++i)
            ply << vertices_[i][0] << " " << vertices_[i][1] << " " << vertices_[i][2] << " " << (int)colors_[i][0] << " " << (int)colors_[i][1] << " " << (int)colors_[i][2] << std::endl;
        for (auto& f : faces_) ply << "3 " << f[0] << " " << f[1] << " " << f[2] << std::endl;
        return true;
    }
};

}  // namespace psgsdf_host
