// Headless render loop: the offscreen branch of the reference's Application::Run
// (application.cpp:58-76) and Application::SaveImage (:157-179), with what a long run without
// a viewer needs on top: progressive images, a film checkpoint and resuming from it.
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ptamd_host.h"
#include "pt_host.h"

namespace pth {

namespace {

constexpr char kMagic[8] = {'P', 'T', 'F', 'I', 'L', 'M', '1', '\0'};

struct CheckpointHeader {
    char magic[8];
    int32_t width, height;
    uint32_t iterations;
    uint32_t reserved;
};

void Check(pt_ctx *ctx, int status, const char *what) {
    if (status != PT_OK) throw Exception(std::string(what) + ": " + pt_last_error(ctx));
}

// Written to a temporary name first: a run killed in the middle of a save keeps its last checkpoint.
void WriteCheckpoint(const std::string &path, int width, int height, uint32_t iterations, const std::vector<float> &film) {
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw Exception("checkpoint: cannot open " + tmp);
    CheckpointHeader h{};
    std::memcpy(h.magic, kMagic, sizeof(kMagic));
    h.width = width;
    h.height = height;
    h.iterations = iterations;
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && std::fwrite(film.data(), sizeof(float), film.size(), f) == film.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        throw Exception("checkpoint: cannot write " + path);
    }
}

}  // namespace

// Returns false when the file does not exist; throws on a malformed file.
bool ReadCheckpoint(const std::string &path, int &width, int &height, uint32_t &iterations, std::vector<float> *film) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    CheckpointHeader h{};
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, kMagic, sizeof(kMagic)) == 0 && h.width > 0 && h.height > 0 &&
              h.width <= 65536 && h.height <= 65536;
    if (ok) {  // the file holds exactly the film its header announces: nothing is allocated on a header's word alone
        const long at = std::ftell(f);
        ok = at >= 0 && std::fseek(f, 0, SEEK_END) == 0;
        const long size = ok ? std::ftell(f) : -1;
        ok = ok && size >= at && uint64_t(size - at) == uint64_t(h.width) * uint64_t(h.height) * 3u * sizeof(float) && std::fseek(f, at, SEEK_SET) == 0;
    }
    if (ok && film) {
        film->resize(size_t(h.width) * h.height * 3);
        ok = std::fread(film->data(), sizeof(float), film->size(), f) == film->size();
    }
    std::fclose(f);
    if (!ok) throw Exception("checkpoint: " + path + " is not a film checkpoint");
    width = h.width;
    height = h.height;
    iterations = h.iterations;
    return true;
}

void RenderOffscreen(pt_ctx *ctx, Scene &scene, const pth_offscreen_options &opt, pth_offscreen_result &result) {
    const int width = scene.Width(), height = scene.Height();
    const uint32_t total = opt.iterations ? opt.iterations : uint32_t(scene.MaxIterations());
    // Samples per pass when the caller leaves it open: a sixteenth of the job, between 8 and 128
    // (beyond 128 the rate no longer grows), and no more than 160 GB of path state: 184 bytes per
    // pixel and sample for each of the three passes in flight.
    uint32_t batch = opt.batch;
    if (batch == 0) {
        const double bytes_per_sample = 3.0 * 184.0 * double(width) * double(height);
        batch = 8;
        while (batch < 128 && batch * 32 <= total && 2.0 * batch * bytes_per_sample <= 160e9) batch *= 2;
    }
    if (total == 0) throw Exception("RenderOffscreen: nothing to render");
    if (batch > 1024) throw Exception("RenderOffscreen: at most 1024 samples per pass");
    Check(ctx, pt_scene_upload(ctx, &scene.GetDeviceScene()), "pt_scene_upload");
    Check(ctx, pt_integrator_init(ctx, width, height), "pt_integrator_init");
    Check(ctx, pt_integrator_reset(ctx, width, height), "pt_integrator_reset");
    Check(ctx, pt_set_render_window(ctx, opt.window[0], opt.window[1], opt.window[2], opt.window[3]), "pt_set_render_window");

    std::vector<float> film;
    uint32_t done = 0;
    if (opt.resume && opt.checkpoint_path) {
        int cw = 0, ch = 0;
        uint32_t ci = 0;
        if (ReadCheckpoint(opt.checkpoint_path, cw, ch, ci, &film)) {
            if (cw != width || ch != height) throw Exception("RenderOffscreen: the checkpoint has another resolution");
            if (ci > 0) {
                Check(ctx, pt_film_upload(ctx, film.data()), "pt_film_upload");
                done = ci;
            }
        }
    }
    result.first_iteration = done;
    result.images_written = 0;
    result.seconds = 0.0;

    std::vector<uint8_t> rgba(size_t(width) * height * 4);
    auto save = [&](uint32_t iterations) {
        if (opt.output_prefix) {
            Check(ctx, pt_display_download(ctx, int(iterations), rgba.data()), "pt_display_download");
            const std::string name = std::string(opt.output_prefix) + "." + std::to_string(iterations) + "samp.png";
            if (pth_save_png(name.c_str(), rgba.data(), width, height) != PT_OK) throw Exception(std::string("pth_save_png: ") + pth_last_error());
            ++result.images_written;
        }
        if (opt.checkpoint_path) {
            film.resize(size_t(width) * height * 3);
            Check(ctx, pt_film_download(ctx, film.data()), "pt_film_download");
            WriteCheckpoint(opt.checkpoint_path, width, height, iterations, film);
        }
    };

    // Passes are submitted back to back (the library keeps several in flight and orders their film
    // updates); the host only waits where it needs the film: at a save point and at the end.
    uint32_t next_save = opt.save_every ? (done / opt.save_every + 1) * opt.save_every : total;
    auto t0 = std::chrono::steady_clock::now();
    while (done < total) {
        // a pass never crosses a save point, so every saved image holds exactly the samples its name says
        const uint32_t stop = next_save < total ? next_save : total;
        const uint32_t samples = (stop - done < batch) ? stop - done : batch;
        // iteration 0 clears the film (path_trace.cu:436-444); a resumed run starts above 0 and keeps it
        Check(ctx, pt_render_batch(ctx, width, height, int(done), int(samples), nullptr), "pt_render_batch");
        done += samples;
        if (done == stop && (done == total || opt.save_every)) {
            Check(ctx, pt_synchronize(ctx), "pt_synchronize");
            result.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            save(done);
            if (opt.save_every) next_save += opt.save_every;
            t0 = std::chrono::steady_clock::now();
        }
    }
    result.iterations_done = done;
}

}  // namespace pth
