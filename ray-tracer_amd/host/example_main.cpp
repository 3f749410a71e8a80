/*
 * example_main.cpp — what the reference's main() (src/main.cu:401-432) looks like on top of
 * raytracer.hpp, minus the SFML window: build a scene with the reference's factories, render
 * progressive frames, write the float->RGBA8 result (src/main.cu:343-371) as a binary PPM or,
 * if the name ends in .png, as a PNG like the reference's images/ directory.
 *
 *   example_main <models_dir> <scene 0..3> <width> <height> <frames> <out.ppm|out.png> [devices, e.g. 0,1,2,3 | pipeN]
 *
 * "pipeN" (N = 1..8) instead of a device list: the loop with N frames in flight on one GPU (submit_frame / collect_frame) -
 * each frame is seeded when it is submitted, as the reference's loop seeds it when it calls render(), and the image is the same.
 *
 * With a device list the same loop runs on a MultiRenderer: every listed GPU renders the bands it owns,
 * the image is the same (a device may be listed twice to rehearse on a one-GPU machine).
 *
 * Build:  g++ -std=c++17 -O2 example_main.cpp -L.. -lraytracer_amd -Wl,-rpath,'$ORIGIN/..'
 */
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "raytracer.hpp"

using namespace rtamd;

int main(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s <models_dir> <scene 0..3> <width> <height> <frames> <out.ppm|out.png> [devices]\n", argv[0]);
        return 2;
    }
    const std::string models = argv[1];
    const int scene_num = std::atoi(argv[2]), W = std::atoi(argv[3]), H = std::atoi(argv[4]), frames = std::atoi(argv[5]);
    try {
        SceneObjects mesh_data(scene_num, models);                 /* init(): src/main.cu:389-398 */
        RenderData render_data(100, 5, true, mesh_data.use_sky ? Vec3(0.8f, 1, 1) : Vec3(0, 0, 0));
        Camera camera(W, H);
        VariableRenderData data{0, std::vector<float>((size_t)W * (size_t)H * 3, 0.0f)};
        std::vector<int> times;
        for (int f = 1; f < frames; f++) times.push_back(12345 + f);
        if (argc > 7 && std::string(argv[7]).compare(0, 4, "pipe") == 0) {
            const int depth = std::atoi(argv[7] + 4) > 0 ? std::atoi(argv[7] + 4) : 4;
            Renderer renderer(0);
            renderer.set_scene(mesh_data);
            renderer.set_frames_in_flight(depth);
            int submitted = 0;
            while (data.frame_num < frames) {
                while (submitted < frames && renderer.frames_in_flight() < depth) renderer.submit_frame(camera, render_data, 12345 + submitted++);   /* get_time() in the reference */
                renderer.collect_frame(&data);                                                           /* ... and the window would draw it here */
            }
            std::printf("%d frames, %d in flight\n", data.frame_num, depth);
        } else if (argc > 7) {
            std::vector<int> devices;
            for (const char *p = argv[7]; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
            MultiRenderer renderer(devices);
            renderer.set_scene(mesh_data);
            renderer.render(camera, render_data, &data, 12345);
            std::printf("frame %d on %d GPUs: %.2f ms\n", data.frame_num, renderer.num_gpus(), renderer.last_kernel_ms());
            renderer.render_frames(camera, render_data, &data, times);
            if (!times.empty()) std::printf("frames 2..%d: %.2f ms\n", data.frame_num, renderer.last_kernel_ms());
        } else {
            Renderer renderer(0);
            renderer.set_scene(mesh_data);
            /* the first frame as the reference does it, one render() per frame ... */
            renderer.render(camera, render_data, &data, 12345);          /* get_time() in the reference */
            std::printf("frame %d: %.2f ms\n", data.frame_num, renderer.last_kernel_ms());
            /* ... the others in one call: the same image, frames overlapping on the GPU */
            renderer.render_frames(camera, render_data, &data, times);
            if (!times.empty()) std::printf("frames 2..%d: %.2f ms\n", data.frame_num, renderer.last_kernel_ms());
        }
        std::vector<uint8_t> rgba = parse_pixel_colours(data.previous_render, W, H);
        const std::string out = argv[6];
        if (out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0) {
            write_png(out, rgba, W, H);
        } else {
            FILE *fp = std::fopen(out.c_str(), "wb");
            if (!fp) throw std::runtime_error("cannot open output file");
            std::fprintf(fp, "P6\n%d %d\n255\n", W, H);
            for (size_t i = 0; i < (size_t)W * (size_t)H; i++) std::fwrite(&rgba[4 * i], 1, 3, fp);
            std::fclose(fp);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
