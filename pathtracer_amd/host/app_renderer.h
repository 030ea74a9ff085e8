// app_renderer.h — the reference's Renderer and Viewer (src/renderer.h:37-71, src/viewer.h) above libptgpu.
//
// Same public methods and the same conventions: non-copyable, non-owning with respect to the caller's Model
// and environment map, std::runtime_error where the reference throws.  The per-ray plugin interfaces of the
// reference (IAccelerationStructure / IRayProgram) are chosen by the enums in RenderSettings, exactly as
// Renderer::load_scene does (src/renderer.cc:44-63); their work runs on the GPU behind ptgpu_render_frame.
#pragma once

#include <cstdint>
#include <filesystem>
#include <memory>
#include <optional>
#include <vector>

#include "app_model.h"
#include "multi_gpu.h"
#include "ptgpu.h"

namespace app {

class Renderer {
  public:
    explicit Renderer(int device = 0);
    // Several GPUs of one node behind the one interface: every device holds the scene, renders its share of each
    // pixel's samples (device k of n: samples k, k + n, ...), and the partial sums are added on the first device with an
    // RCCL reduce before the mean is taken.  `always_reduce` sends a single device through the same path (a
    // communicator of one).  Not in the reference, whose frame loop runs on one CPU.
    Renderer(const std::vector<int>& devices, bool always_reduce);
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    ~Renderer();

    void update_camera_transform_state(fvec3 position, fvec3 direction, fvec3 up, Perspective perspectiveParams);
    void load_scene(const Model& model, const CPUTexture<hdr_pixel>& envmap);
    void render_frame(CPUFrameBuffer& framebuffer);
    void set_render_settings(const RenderSettings& settings);
    RenderSettings get_render_settings() const;
    BBox get_scene_bound() const;

    // additions of the GPU build
    void set_seed(uint32_t seed) { seed_ = seed; push_settings(); }
    void set_single_path(bool on) { single_path_ = on; push_settings(); }  // PTGPU_FLAG_SINGLE_PATH
    void set_wide_nodes(bool on) { wide_nodes_ = on; push_settings(); }    // PTGPU_FLAG_WIDE_NODES, takes effect at load_scene
    const ptgpu_stats& last_stats() const { return stats_; }
    // closest hit of sample `sample` of every pixel, row-major (the raycast parity dump)
    std::vector<ptgpu_hit> render_hits(int width, int height, uint32_t sample = 0);

  private:
    void push_settings();
    [[noreturn]] void fail(const char* what, int code) const;

    void render_frame_on_group(CPUFrameBuffer& framebuffer);
    ptgpu_ctx* ctx_ = nullptr;           // the first (or only) device's context
    std::vector<ptgpu_ctx*> ctxs_;       // one per device, ctxs_[0] == ctx_
    std::unique_ptr<DeviceGroup> group_; // set when frames go through the reduce
    RenderSettings renderSettings_;
    uint32_t seed_ = 0;
    bool single_path_ = false, wide_nodes_ = false;
    ptgpu_stats stats_{};
};

class Viewer {  // src/viewer.h: owns model, environment, renderer and framebuffer
  public:
    Viewer(Model&& model, CPUTexture<hdr_pixel>&& environmentTexture, const RenderSettings& settings, int device = 0,
           bool wide_nodes = false);  // wide_nodes: Renderer::set_wide_nodes before the scene is loaded
    Viewer(Model&& model, CPUTexture<hdr_pixel>&& environmentTexture, const RenderSettings& settings, const std::vector<int>& devices,
           bool always_reduce, bool wide_nodes);  // several GPUs, see Renderer
    void resize_window(const ivec2& newDimensions);
    ivec2 get_window_dimensions() const { return windowDimensions_; }
    void render() { renderer_.render_frame(framebuffer_); }
    void set_active_camera(std::optional<uint32_t> cameraIndex);
    std::optional<uint32_t> get_active_camera() const { return activeCameraIndex_; }
    void take_snapshot(const std::filesystem::path& filePath) const { framebuffer_.save_to_file(filePath); }
    bool snap_to_camera();
    void set_render_settings(const RenderSettings& settings);
    RenderSettings get_render_settings() const { return renderer_.get_render_settings(); }
    Renderer& renderer() { return renderer_; }
    const CPUFrameBuffer& framebuffer() const { return framebuffer_; }

  private:
    void start(const RenderSettings& settings, bool wide_nodes);
    void set_up_default_camera_transforms();

    Model model_;
    CPUTexture<hdr_pixel> environmentTexture_;
    Renderer renderer_;
    CPUFrameBuffer framebuffer_;
    ivec2 windowDimensions_{800, 600};
    std::optional<uint32_t> activeCameraIndex_;
    fvec3 position_{0, 0, 0}, direction_{0, 0, -1}, up_{0, 1, 0};
    Perspective cam_params_;
};

}  // namespace app
