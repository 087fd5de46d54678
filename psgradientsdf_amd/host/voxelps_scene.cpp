// voxelps_scene — drives the C++ mirror classes exactly the way main_ps.cpp:193-202,323-330 drives the reference's,
// on a scene dumped as raw little-endian arrays (tests write it with numpy; the PNG / pose-file loaders of the
// reference are SURVEY §8f "next" rows).  Usage: voxelps_scene <scene_dir>/ <output_dir>/
//   scene_dir/meta.txt : dim0 dim1 dim2 voxel_size shift0 shift1 shift2 T F W H vis_words model(0|1|2) max_it conv upsample reg_n reg_l damping
//   K.f32 (9) dist.f32 grad.f32 weight.f32 rgb.f32 vis.u64 images.f32 poses.f32 frame_idx.i32
#include <cstdio>
#include <cstdlib>

#include "ps_optimizer.hpp"

using namespace psgsdf_host;

template <class T> static bool slurp(const std::string& path, std::vector<T>& v, size_t n) {
    v.resize(n);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { std::cerr << "cannot open " << path << std::endl; return false; }
    size_t got = fread(v.data(), sizeof(T), n, f);
    fclose(f);
    return got == n;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: voxelps_scene <scene_dir>/ <output_dir>/" << std::endl; return 1; }
    std::string in = argv[1], out = argv[2];
    std::ifstream meta(in + "meta.txt");
    if (!meta.is_open()) { std::cerr << "no meta.txt in " << in << std::endl; return 1; }
    VolumetricGradSdf vol; int F, W, H, model, max_it, upsample; float conv, reg_n, reg_l, damping;
    meta >> vol.grid_dim_[0] >> vol.grid_dim_[1] >> vol.grid_dim_[2] >> vol.voxel_size_ >> vol.shift_[0] >> vol.shift_[1] >> vol.shift_[2] >> vol.T_
         >> F >> W >> H >> vol.vis_words >> model >> max_it >> conv >> upsample >> reg_n >> reg_l >> damping;
    const size_t n = vol.num_voxels();
    std::vector<float> K, images, poses; std::vector<int> fidx;
    if (!slurp(in + "K.f32", K, 9) || !slurp(in + "dist.f32", vol.dist, n) || !slurp(in + "grad.f32", vol.grad, 3 * n) || !slurp(in + "weight.f32", vol.weight, n) ||
        !slurp(in + "rgb.f32", vol.rgb, 3 * n) || !slurp(in + "vis.u64", vol.vis, n * vol.vis_words) || !slurp(in + "images.f32", images, (size_t)F * W * H * 3) ||
        !slurp(in + "poses.f32", poses, (size_t)F * 16) || !slurp(in + "frame_idx.i32", fidx, F)) return 1;
    OptimizerSettings* opt_set_ = new OptimizerSettings();
    opt_set_->model = (ModelType)model; opt_set_->order = model == 1 ? 2 : 1; opt_set_->max_it = max_it; opt_set_->conv_threshold = conv; opt_set_->upsample = upsample != 0;
    opt_set_->reg_weight_n = reg_n; opt_set_->reg_weight_l = reg_l; opt_set_->damping = damping; opt_set_->lambda = 0.2f; opt_set_->lambda_sq = 0.04f;
    Mat3f Km; for (int i = 0; i < 9; ++i) Km.v[i] = K[i];

    // ---- from here on: the call sequence of main_ps.cpp:193-202,323-330
    Optimizer* vOpt;
    switch (opt_set_->model) {
        case SH1: case SH2: vOpt = new PsOptimizer(&vol, vol.voxel_size_, Km, out, opt_set_); break;
        default: vOpt = new LedOptimizer(&vol, vol.voxel_size_, Km, out, opt_set_); break;
    }
    std::vector<std::shared_ptr<ImageRGB>> key_images; std::vector<Mat4f> key_poses; std::vector<std::string> key_stamps;
    for (int f = 0; f < F; ++f) {
        auto im = std::make_shared<ImageRGB>(); im->rows = H; im->cols = W;
        im->data.assign(images.begin() + (size_t)f * W * H * 3, images.begin() + (size_t)(f + 1) * W * H * 3);
        key_images.push_back(im);
        Mat4f P; std::copy(poses.begin() + 16 * f, poses.begin() + 16 * f + 16, P.begin()); key_poses.push_back(P);
        key_stamps.push_back(std::to_string(fidx[f]));
    }
    vOpt->setImages(key_images);
    vOpt->setKeyframes(fidx);
    vOpt->setKeytimestamps(key_stamps);
    vOpt->setPoses(key_poses);
    vOpt->init();
    bool ok = vOpt->alternatingOptimize(true, true, true, true);
    vOpt->savePoses("final_poses");
    // dump the refined distances for the parity test
    FILE* f = fopen((out + "dist_out.f32").c_str(), "wb");
    if (f) { fwrite(vol.dist.data(), sizeof(float), vol.dist.size(), f); fclose(f); }
    std::cout << "alternatingOptimize returned " << ok << std::endl;
    delete vOpt; delete opt_set_;
    return 0;
}
