// image_loader.hpp — dataset readers of img_loader/{ImageLoader,SynthLoader,MultiviewLoader,TumrgbdLoader}.h without OpenCV:
// intrinsics (9 floats), depth PNG16 * unit -> float metres, colour PNG8 -> float RGB / 255 (the reference keeps BGR, the
// engine wants RGB), TUM-format pose file "stamp tx ty tz qx qy qz qw".
#pragma once
#include <deque>
#include <future>
#include <iomanip>
#include <sstream>

#include "png_reader.hpp"
#include "ps_optimizer.hpp"

namespace psgsdf_host {

struct DepthImage { int rows = 0, cols = 0; std::vector<float> data; };

class ImageLoader {
protected:
    Mat3f K_{};
    const float unit_;
    const std::string path_;
    std::string timestamp_rgb_, timestamp_depth_;
public:
    ImageLoader(float unit, const std::string& path) : unit_(unit), path_(path) {}
    virtual ~ImageLoader() {}
    Mat3f K() const { return K_; }
    std::string rgb_timestamp() { return timestamp_rgb_; }
    std::string depth_timestamp() { return timestamp_depth_; }
    // ImageLoader.h:109-128: the first 9 numbers of the file
    bool load_intrinsics(const std::string& filename = "intrinsics.txt") {
        std::ifstream in(path_ + filename);
        if (!in.is_open()) return false;
        for (int i = 0; i < 9; ++i) { float t = 0; in >> t; K_.v[i] = t; }
        return true;
    }
    // ImageLoader.h:130-146: IMREAD_ANYDEPTH, convertTo(CV_32FC1, unit)
    bool load_depth(const std::string& filename, DepthImage& depth) {
        PngImage p;
        if (!read_png(path_ + filename, p)) { std::cerr << "Error: empty depth image " << path_ + filename << std::endl; return false; }
        depth.rows = p.height; depth.cols = p.width; depth.data.resize((size_t)p.width * p.height);
        for (size_t i = 0; i < depth.data.size(); ++i) depth.data[i] = (float)p.px[i * p.channels] * unit_;
        return true;
    }
    // ImageLoader.h:167-188: imread (8-bit, 3 channels), convertTo(CV_32FC3, 1/255)
    bool load_color(const std::string& filename, ImageRGB& color) {
        PngImage p;
        if (!read_png(path_ + filename, p)) { std::cerr << "Error: empty color image " << path_ + filename << std::endl; return false; }
        color.rows = p.height; color.cols = p.width; color.data.resize((size_t)p.width * p.height * 3);
        const int sh = p.bit_depth == 16 ? 8 : 0;
        for (size_t i = 0; i < (size_t)p.width * p.height; ++i)
            for (int c = 0; c < 3; ++c) { int src = p.channels >= 3 ? c : 0; color.data[3 * i + c] = (float)(p.px[i * p.channels + src] >> sh) * (1.0f / 255.0f); }
        return true;
    }
    virtual bool load_next(ImageRGB& color, DepthImage& depth) = 0;
    virtual void reset_counter() {}
    // The two halves of load_next: naming the next frame's files (cheap, sequential, sets the time stamps) and decoding them (expensive, touches no
    // loader state: safe on worker threads).  FramePrefetcher below decodes several frames ahead of the fusion.
    virtual bool next_names(std::string& depth_fn, std::string& rgb_fn) = 0;
    bool decode(const std::string& depth_fn, const std::string& rgb_fn, ImageRGB& color, DepthImage& depth) { return load_depth(depth_fn, depth) && load_color(rgb_fn, color); }
    // ImageLoader.h:228-258 (Eigen::Quaternionf::toRotationMatrix)
    bool load_pose(const std::string& filename, std::vector<Mat4f>& poses) {
        std::ifstream file((path_ + filename).c_str());
        if (!file.is_open()) { std::cout << "can't load poses!" << std::endl; return false; }
        std::string line;
        while (std::getline(file, line)) {
            float ts, t[3], qx, qy, qz, qw; std::stringstream s(line);
            if (!(s >> ts >> t[0] >> t[1] >> t[2] >> qx >> qy >> qz >> qw)) continue;
            if (qw * qw + qx * qx + qy * qy + qz * qz < 0.99) std::cerr << "pose " << ts << " has invalid rotation" << std::endl;
            const float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            Mat4f M = {1 - (tyy + tzz), txy - twz, txz + twy, t[0], txy + twz, 1 - (txx + tzz), tyz - twx, t[1], txz - twy, tyz + twx, 1 - (txx + tyy), t[2], 0, 0, 0, 1};
            poses.push_back(M);
        }
        return !poses.empty();
    }
};

// SynthLoader.h:35-57: depth/NNN.png, rgb/NNN.png, unit 1/1000, counter from 1
class SynthLoader : public ImageLoader {
    size_t counter = 1;
public:
    explicit SynthLoader(const std::string& path) : ImageLoader(1.f / 1000, path) {}
    bool next_names(std::string& depth_fn, std::string& rgb_fn) override {
        std::stringstream ss; ss << std::setfill('0') << std::setw(3) << counter;
        timestamp_rgb_ = ss.str(); timestamp_depth_ = timestamp_rgb_;
        depth_fn = "depth/" + timestamp_rgb_ + ".png"; rgb_fn = "rgb/" + timestamp_rgb_ + ".png";
        ++counter; return true;
    }
    bool load_next(ImageRGB& color, DepthImage& depth) override {
        std::string d, r; next_names(d, r);
        if (!load_depth(d, depth) || !load_color(r, color)) { --counter; return false; }      // (the reference only counts a frame it could load)
        return true;
    }
    void reset_counter() override { counter = 1; }
};
// MultiviewLoader.h:35-58: depthNNNNNN.png, colorNNNNNN.png, unit 1/1000, counter from 1
class MultiviewLoader : public ImageLoader {
    size_t counter = 1;
public:
    explicit MultiviewLoader(const std::string& path) : ImageLoader(1.f / 1000, path) {}
    bool next_names(std::string& depth_fn, std::string& rgb_fn) override {
        std::stringstream ss; ss << std::setfill('0') << std::setw(6) << counter;
        timestamp_rgb_ = ss.str(); timestamp_depth_ = timestamp_rgb_;
        depth_fn = "depth" + timestamp_rgb_ + ".png"; rgb_fn = "color" + timestamp_rgb_ + ".png";
        ++counter; return true;
    }
    bool load_next(ImageRGB& color, DepthImage& depth) override {
        std::string d, r; next_names(d, r);
        if (!load_depth(d, depth) || !load_color(r, color)) { --counter; return false; }
        return true;
    }
    void reset_counter() override { counter = 1; }
};
// TumrgbdLoader.h:83-119: associated.txt "rgb_stamp rgb_file depth_stamp depth_file", unit 1/5000; reset_counter is a no-op (B12)
class TumrgbdLoader : public ImageLoader {
    std::ifstream assoc_;
public:
    explicit TumrgbdLoader(const std::string& path) : ImageLoader(1.f / 5000, path) { assoc_.open(path_ + "associated.txt"); }
    bool next_names(std::string& depth_fn, std::string& rgb_fn) override {
        std::string line = "#";
        while (line.empty() || line.at(0) == '#') if (!std::getline(assoc_, line)) return false;
        std::istringstream ss(line); ss >> timestamp_rgb_ >> rgb_fn >> timestamp_depth_ >> depth_fn;
        return true;
    }
    bool load_next(ImageRGB& color, DepthImage& depth) override {
        std::string d, r;
        if (!next_names(d, r)) return false;
        if (!load_depth(d, depth)) return false;
        return load_color(r, color);
    }
};

// Decodes the PNGs of the next few frames on worker threads while the current frame is being fused / tracked (a 1139 x 1709
// colour + depth pair takes ~50 ms of single-threaded inflate + unfilter; the fusion of a frame takes a few ms).  Frames come out in sequence order
// with their time stamps; a frame that cannot be read ends the sequence there, exactly as load_next would have.
class FramePrefetcher {
public:
    struct Frame { bool ok = false; ImageRGB color; DepthImage depth; std::string rgb_stamp, depth_stamp; };
private:
    ImageLoader* l_; size_t window_; bool exhausted_ = false;
    std::deque<std::future<std::shared_ptr<Frame>>> q_;
    void fill() {
        while (!exhausted_ && q_.size() < window_) {
            std::string d, r;
            if (!l_->next_names(d, r)) { exhausted_ = true; break; }
            const std::string rs = l_->rgb_timestamp(), ds = l_->depth_timestamp();
            ImageLoader* l = l_;
            q_.push_back(std::async(std::launch::async, [l, d, r, rs, ds] {
                auto f = std::make_shared<Frame>(); f->rgb_stamp = rs; f->depth_stamp = ds;
                // the two PNGs of a frame side by side: the first frame of a run is not hidden behind anything (1139 x 1709: 60 ms -> 35 ms)
                auto dep = std::async(std::launch::async, [&] { return l->load_depth(d, f->depth); });
                const bool col = l->load_color(r, f->color);
                f->ok = dep.get() && col;
                return f; }));
        }
    }
public:
    FramePrefetcher(ImageLoader* l, size_t window) : l_(l), window_(std::max<size_t>(1, window)) {}
    // the next frame of the sequence, or null when the files are exhausted / unreadable
    std::shared_ptr<Frame> next() {
        fill();
        if (q_.empty()) return nullptr;
        auto f = q_.front().get(); q_.pop_front();
        if (!f->ok) { exhausted_ = true; q_.clear(); return nullptr; }
        fill();
        return f;
    }
};

// SharpDetector.h:12-37 modifiedLaplacian on the colour image: sepFilter2D with M = [-1 2 -1], G = getGaussianKernel(3,-1) =
// [1/4 1/2 1/4], BORDER_REFLECT_101; cv::mean(FM).val[0] = mean of channel 0, which is BLUE in the reference's BGR images.
inline float modifiedLaplacian(const ImageRGB& img) {
    const int H = img.rows, W = img.cols; const int ch = 2;   // blue
    auto rx = [&](int x) { return x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x); };
    auto ry = [&](int y) { return y < 0 ? -y : (y >= H ? 2 * H - 2 - y : y); };
    // separable like cv::sepFilter2D: the row kernel first (float), then the column kernel over the row results
    std::vector<float> rowM((size_t)W * H), rowG((size_t)W * H);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const float a = img.data[((size_t)y * W + rx(x - 1)) * 3 + ch], b = img.data[((size_t)y * W + x) * 3 + ch], c = img.data[((size_t)y * W + rx(x + 1)) * 3 + ch];
        rowM[(size_t)y * W + x] = (-a + 2.0f * b) - c;
        rowG[(size_t)y * W + x] = (0.25f * a + 0.5f * b) + 0.25f * c;
    }
    double sum = 0;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const size_t u = (size_t)ry(y - 1) * W + x, m = (size_t)y * W + x, d = (size_t)ry(y + 1) * W + x;
        const float lx = (0.25f * rowM[u] + 0.5f * rowM[m]) + 0.25f * rowM[d];      // Lx = sepFilter2D(src, M, G): M along x, G along y
        const float ly = (-rowG[u] + 2.0f * rowG[m]) - rowG[d];                      // Ly = sepFilter2D(src, G, M)
        sum += (double)(std::fabs(lx) + std::fabs(ly));
    }
    return (float)(sum / ((double)W * H));
}
inline bool sharpDetector(const ImageRGB& img, float threshold) {
    float measure = modifiedLaplacian(img);
    std::cout << "======> the sharpness measure is " << measure << "." << std::endl;
    return !(measure < threshold);
}

}  // namespace psgsdf_host
