// PathTracer — command line of the reference (src/main.cc:12-62, src/arguments.cc:28-147) above the GPU renderer.
// Same flags, defaults, stdout lines and exit codes; -f takes glTF 2.0 (.glb / .gltf) or the flat scene container.
// Additive flags: --gpus N (sample split over N devices of this node + RCCL reduce), --collective (the reduce path even with
// one device), --seed, --device, --dump-hits FILE (closest hit of sample 0 per pixel, raw ptgpu_hit records,
// row-major), --stats, -a 3 (per-mesh trees built on the GPU), --wide-nodes (four-wide quantised nodes) and --single-path (PBR keeps one ray per hit, chosen in proportion to its payload).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "app_renderer.h"

namespace {

struct Args {
    std::string model, env = "white", output = "snapshot.hdr", dump_hits;
    int program = 3, as = 2, samples = 1, bounces = 4, newrays = 0, env_rot = 0, max_leaf = 8, width = 800, height = 600;
    unsigned seed = 0;
    int device = 0, gpus = 1;
    bool stats = false, single_path = false, wide_nodes = false, collective = false;
};

const char* kUsage =
    "Usage: PathTracer -f <scene.glb|.gltf|.ptsc> [-e <env.hdr>|black|white] [-o <out.png|out.hdr>] [-p 1|2|3] [-a 1|2|3]\n"
    "                  [-s samples] [-b bounces] [-r newrays] [--env_rot degrees] [-m max_leaf_triangles]\n"
    "                  [--width W] [--height H] [--seed N] [--device N] [--gpus N] [--collective]\n"
    "                  [--dump-hits FILE] [--stats] [--single-path] [--wide-nodes]\n"
    "Use -h or --help for usage info.\n";

bool parse(int argc, char** argv, Args& a) {
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) {
            std::cerr << "missing value after " << argv[i] << "\n" << kUsage;
            return nullptr;
        }
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        const char* v = nullptr;
        auto num = [&](int& dst) { if (!(v = need(i))) return false; dst = std::atoi(v); return true; };
        if (k == "-h" || k == "--help") { std::cout << kUsage; return false; }
        else if (k == "-f" || k == "--file") { if (!(v = need(i))) return false; a.model = v; }
        else if (k == "-e" || k == "--env") { if (!(v = need(i))) return false; a.env = v; }
        else if (k == "-o" || k == "--output") { if (!(v = need(i))) return false; a.output = v; }
        else if (k == "--dump-hits") { if (!(v = need(i))) return false; a.dump_hits = v; }
        else if (k == "-p" || k == "--program") { if (!num(a.program)) return false; }
        else if (k == "-a" || k == "--as") { if (!num(a.as)) return false; }
        else if (k == "-s" || k == "--samples") { if (!num(a.samples)) return false; }
        else if (k == "-b" || k == "--bounces") { if (!num(a.bounces)) return false; }
        else if (k == "-r" || k == "--newrays") { if (!num(a.newrays)) return false; }
        else if (k == "--env_rot") { if (!num(a.env_rot)) return false; }
        else if (k == "-m") { if (!num(a.max_leaf)) return false; }
        else if (k == "--width") { if (!num(a.width)) return false; }
        else if (k == "--height") { if (!num(a.height)) return false; }
        else if (k == "--device") { if (!num(a.device)) return false; }
        else if (k == "--gpus") { if (!num(a.gpus)) return false; }
        else if (k == "--collective") { a.collective = true; }
        else if (k == "--seed") { if (!(v = need(i))) return false; a.seed = static_cast<unsigned>(std::strtoul(v, nullptr, 10)); }
        else if (k == "--stats") { a.stats = true; }
        else if (k == "--single-path") { a.single_path = true; }
        else if (k == "--wide-nodes") { a.wide_nodes = true; }
        else { std::cerr << "unknown argument " << k << "\n" << kUsage; return false; }
    }
    // Out-of-range values are repaired, not rejected, exactly as the reference does: an enum outside its range becomes
    // its first value (int_to_enum, src/arguments.cc:14-18: -p 9 is the ray caster, -a 0 the naive structure), counts are
    // clamped from below (src/arguments.cc:121-132).
    if (a.program < 1 || a.program > 3) a.program = 1;
    if (a.as < 1 || a.as > 3) a.as = 1;
    if (a.samples < 1) a.samples = 1;
    if (a.bounces < 0) a.bounces = 0;
    if (a.newrays < 0) a.newrays = 0;
    if (a.max_leaf < 1) a.max_leaf = 1;
    if (a.width < 1) a.width = 1;
    if (a.height < 1) a.height = 1;
    if (a.gpus < 1) a.gpus = 1;
    if (a.model.empty()) {  // src/arguments.cc:136-139: the message, then exit code 1 (src/main.cc:18-20)
        std::cout << "A PathTracer for .gltf files on MI355X.\nUse -h or --help for usage info." << std::endl;
        return false;
    }
    return true;
}

long ms_since(std::chrono::high_resolution_clock::time_point t0) {
    return static_cast<long>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count());
}

}  // namespace

int main(int argc, char* argv[]) {
    using namespace app;
    Args args;
    if (!parse(argc, argv, args)) return 1;
    try {
        auto start = std::chrono::high_resolution_clock::now();
        Model model;
        {
            ModelLoader loader{};
            if (!loader.loadFromFile(args.model)) {
                std::cerr << "Failed to load model from " << args.model << '\n';
                return 1;
            }
            model = loader.constructModel();
        }
        CPUTexture<hdr_pixel> environment_texture;
        // an empty path is the default environment too, and the default is white (src/arguments.cc:140-144, arguments.h:32)
        if (args.env == "white" || args.env.empty()) environment_texture = CPUTexture<hdr_pixel>::create_white_texture();
        else if (args.env == "black") environment_texture = CPUTexture<hdr_pixel>::create_black_texture();
        else environment_texture = CPUTexture<hdr_pixel>(std::filesystem::path(args.env));
        std::cout << "loaded in " << ms_since(start) << " ms." << '\n';

        RenderSettings settings;  // RenderSettings(const ConsoleArgs&), src/render_settings.cc:5-14
        settings.samplesPerPixel = static_cast<unsigned>(args.samples);
        settings.maxRayBounces = static_cast<unsigned>(args.bounces);
        settings.maxNewRaysPerBounce = static_cast<unsigned>(args.newrays);
        settings.maxTrianglesPerBVHLeaf = static_cast<unsigned>(args.max_leaf);
        settings.envmapRotation = static_cast<float>(args.env_rot) * 0.01745329251994329576923690768489f;
        settings.programMode = static_cast<RayProgramMode>(args.program - 1);
        settings.accelStructType = static_cast<AccelerationStructureType>(args.as - 1);

        // --gpus N: devices --device, --device + 1, ...; each renders its share of every pixel's samples, the partial sums
        // are reduced onto the first one over RCCL (--collective: also with one device, a communicator of one)
        std::vector<int> devices;
        for (int g = 0; g < args.gpus; ++g) devices.push_back(args.device + g);
        const bool grouped = args.gpus > 1 || args.collective;
        Viewer viewer = grouped ? Viewer(std::move(model), std::move(environment_texture), settings, devices, args.collective, args.wide_nodes)
                                : Viewer(std::move(model), std::move(environment_texture), settings, args.device, args.wide_nodes);
        viewer.renderer().set_seed(args.seed);
        viewer.renderer().set_single_path(args.single_path);
        viewer.resize_window(ivec2{args.width, args.height});
        viewer.snap_to_camera();

        start = std::chrono::high_resolution_clock::now();
        viewer.render();
        std::cout << "rendered in " << ms_since(start) << " ms." << '\n';
        if (args.stats) {
            const ptgpu_stats& s = viewer.renderer().last_stats();
            std::cout << "rays " << s.rays << " samples " << s.camera_samples << " gpu_ms " << s.total_ms << " extend_ms " << s.extend_ms
                      << " shade_ms " << s.shade_ms << " Mrays/s " << (s.total_ms > 0 ? s.rays / s.total_ms / 1e3 : 0.0) << '\n';
            // (one, however many devices render: the first builds, the others copy its device buffers)
            std::cout << "acceleration structure builds " << ptgpu_host_build_count() << " devices " << args.gpus << '\n';
            if (s.rays_abandoned)  // (flipped vertex normals under an alpha-blended material, mostly: see ptgpu_stats)
                std::cout << "pass-through chains cut because they did not advance: " << s.rays_abandoned << '\n';
        }

        start = std::chrono::high_resolution_clock::now();
        viewer.take_snapshot(args.output);
        if (!args.dump_hits.empty()) {
            const std::vector<ptgpu_hit> hits = viewer.renderer().render_hits(args.width, args.height, 0);
            std::ofstream f(args.dump_hits, std::ios::binary);
            f.write(reinterpret_cast<const char*>(hits.data()), static_cast<std::streamsize>(hits.size() * sizeof(ptgpu_hit)));
        }
        std::cout << "output saved in " << ms_since(start) << " ms." << '\n';
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << '\n';
        return 1;
    }
    return 0;
}
