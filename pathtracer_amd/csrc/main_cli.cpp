// Headless command line: `pathtracer <conf> [overrides]`, the process-level contract of the
// reference binary (src/main.cpp:28-53 of the reference: exactly one conf argument, paths inside
// the conf relative to the working directory, "[error] ..." on stdout and exit code 1 on failure,
// ASCII PPM written to PathTracer.ResultPath).
#include "../../include/ptx.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int die(const char *what)
{
    std::printf("[error] %s: %s\n", what, ptx_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2)
    {
        std::printf("[error] usage: pathtracer <conf> [--width W] [--height H] [--spp N] [--depth D] [--out PATH] "
                    "[--mode f64|f32] [--estimator reference|fast] [--gpu ID] [--seed S] [--root DIR] [--model OBJ]\n"
                    "        several GPUs of this node: [--gpus N | --devices ID,ID,...] [--scheme bands|samples] "
                    "[--exchange rccl|peer]\n"
                    "        a part of the frame for later merging (python -m pathtracer_amd.image --merge): "
                    "--dump-partial FILE [--rows A:B] [--samples A:B] [--resume: skip the part if FILE already holds it]\n");
        return 1;
    }
    ptx_overrides ov;
    std::memset(&ov, 0, sizeof(ov));
    ptx_render_opts ro;
    std::memset(&ro, 0, sizeof(ro));
    ro.struct_size = sizeof(ro);
    ro.seed = 13;
    std::string out_path, root, dump_path;
    int rows_a = 0, rows_b = 0, smp_a = 0, smp_b = 0; // --rows A:B, --samples A:B (half-open; 0:0 = all)
    bool resume = false;                              // --resume: a part file that is already complete is left alone
    auto range = [](const char *txt, int &a, int &b) {
        const char *colon = std::strchr(txt, ':');
        a = std::atoi(txt);
        b = colon ? std::atoi(colon + 1) : 0;
        return colon != nullptr && a >= 0 && b > a;
    };
    int gpu = 0;
    std::vector<int> devices; // more than one entry: the frame is sharded over these GPUs
    int scheme = PTX_SHARD_BANDS, exchange = PTX_EXCHANGE_RCCL;
    for (int i = 2; i < argc; i++)
    {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--width")
            ov.width = std::atoi(next());
        else if (a == "--height")
            ov.height = std::atoi(next());
        else if (a == "--spp")
            ov.spp = std::atoi(next());
        else if (a == "--depth")
            ov.depth = std::atoi(next());
        else if (a == "--out")
            out_path = next();
        else if (a == "--root")
            root = next();
        else if (a == "--model")
            ov.model = next();
        else if (a == "--gpu")
            gpu = std::atoi(next());
        else if (a == "--gpus")
        {
            devices.clear();
            for (int k = 0, n = std::atoi(next()); k < n; k++)
                devices.push_back(k);
        }
        else if (a == "--devices")
        {
            devices.clear();
            std::string list = next();
            for (size_t p = 0; p < list.size();)
            {
                size_t q = list.find(',', p);
                if (q == std::string::npos)
                    q = list.size();
                devices.push_back(std::atoi(list.substr(p, q - p).c_str()));
                p = q + 1;
            }
        }
        else if (a == "--scheme")
            scheme = std::string(next()) == "samples" ? PTX_SHARD_SAMPLES : PTX_SHARD_BANDS;
        else if (a == "--exchange")
            exchange = std::string(next()) == "peer" ? PTX_EXCHANGE_PEER : PTX_EXCHANGE_RCCL;
        else if (a == "--seed")
            ro.seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--mode")
            ro.mode = std::string(next()) == "f32" ? PTX_MODE_F32 : PTX_MODE_F64;
        else if (a == "--estimator")
            ro.estimator = std::string(next()) == "fast" ? PTX_ESTIMATOR_FAST : PTX_ESTIMATOR_REFERENCE;
        else if (a == "--dump-partial")
            dump_path = next();
        else if (a == "--resume")
            resume = true;
        else if (a == "--rows" || a == "--samples")
        {
            int &lo = a == "--rows" ? rows_a : smp_a, &hi = a == "--rows" ? rows_b : smp_b;
            if (!range(next(), lo, hi))
            {
                std::printf("[error] %s takes A:B with 0 <= A < B\n", a.c_str());
                return 1;
            }
        }
        else
        {
            std::printf("[error] unknown option %s\n", a.c_str());
            return 1;
        }
    }
    ptx_scene *scene = nullptr;
    if (ptx_scene_load_conf(argv[1], root.empty() ? nullptr : root.c_str(), &ov, &scene))
        return die("load conf");
    ptx_scene_info info;
    ptx_scene_get_info(scene, &info);
    std::printf("[log] scene: %d faces, %d materials, %d lights, %dx%d, %d spp, depth %d\n", info.n_tris, info.n_mats,
                info.n_lights, info.width, info.height, info.spp, info.depth);
    if ((rows_b || smp_b) && dump_path.empty())
    {
        std::printf("[error] --rows / --samples render a part of the frame: name the file it goes to with --dump-partial\n");
        return 1;
    }
    if (!dump_path.empty())
    {
        // a part: rows [A, B) of the frame and samples [A, B) of the conf's Samples, whose count stays the divisor
        if (rows_b > info.height || smp_b > info.spp || devices.size() > 1)
        {
            std::printf("[error] --dump-partial: rows beyond the frame, samples beyond the scene's %d, or more than one GPU\n", info.spp);
            return 1;
        }
        if (!rows_b)
            rows_a = 0, rows_b = info.height;
        if (!smp_b)
            smp_a = 0, smp_b = info.spp;
        ro.row_begin = rows_a, ro.row_end = rows_b;
        ro.sample_offset = smp_a, ro.spp = smp_b - smp_a, ro.spp_total = info.spp;
        if (resume)
        {
            // a run that was cut short starts again with the same command lines: a part whose file is there,
            // names this very piece of this very frame and has its full length is not rendered again
            const int32_t want[12] = {info.width, info.height, info.spp, rows_a, rows_b, smp_a, smp_b, ro.mode, ro.estimator, info.depth, 0, 0};
            if (FILE *f = std::fopen(dump_path.c_str(), "rb"))
            {
                char magic[8];
                int32_t head[12];
                uint64_t seed = 0;
                bool same = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTXPART1", 8) == 0 && std::fread(head, sizeof(head), 1, f) == 1 &&
                            std::memcmp(head, want, sizeof(want)) == 0 && std::fread(&seed, 8, 1, f) == 1 && seed == ro.seed;
                if (same)
                {
                    std::fseek(f, 0, SEEK_END);
                    same = std::ftell(f) == 64 + static_cast<long>(rows_b - rows_a) * info.width * 3 * 8;
                }
                std::fclose(f);
                if (same)
                {
                    std::printf("[debug] %s already holds rows %d:%d, samples %d:%d: nothing to do\n", dump_path.c_str(), rows_a, rows_b, smp_a, smp_b);
                    ptx_scene_destroy(scene);
                    return 0;
                }
            }
        }
    }
    else if (resume)
    {
        std::printf("[error] --resume goes with --dump-partial\n");
        return 1;
    }
    if (devices.empty())
        devices.push_back(gpu);
    // one context per GPU, each with its own copy of the scene
    std::vector<ptx_context *> ctxs(devices.size(), nullptr);
    for (size_t k = 0; k < devices.size(); k++)
    {
        if (ptx_context_create(devices[k], &ctxs[k]))
            return die("create context");
        if (ptx_build_accel(ctxs[k], scene))
            return die("build accel");
    }
    ptx_context *ctx = ctxs[0];
    std::vector<double> sum(static_cast<size_t>(info.width) * info.height * 3), img(sum.size());
    ptx_stats st;
    std::memset(&st, 0, sizeof(st));
    st.struct_size = sizeof(st);
    if (ctxs.size() == 1)
    {
        if (ptx_render(ctx, &ro, sum.data(), 0, &st))
            return die("render");
    }
    else
    {
        // interleaved row bands of 8 (or sample ranges), one host thread per GPU inside the library,
        // then the one exchange: the per-pixel sums reduced onto the first GPU
        std::vector<ptx_stats> per(ctxs.size());
        std::memset(per.data(), 0, per.size() * sizeof(ptx_stats));
        per[0].struct_size = sizeof(ptx_stats);
        if (ptx_render_sharded(ctxs.data(), static_cast<int32_t>(ctxs.size()), &ro, scheme, exchange, sum.data(), per.data()))
            return die("sharded render");
        for (size_t k = 0; k < per.size(); k++)
        {
            std::printf("[log] gpu %d: %llu paths, %.3f ms\n", devices[k], (unsigned long long)per[k].paths, per[k].ms_total);
            st.paths += per[k].paths, st.segments += per[k].segments, st.shadow_rays += per[k].shadow_rays;
            st.ms_total = per[k].ms_total > st.ms_total ? per[k].ms_total : st.ms_total; // the GPUs run side by side
        }
    }
    std::printf("[log] cPathTracer::RayCast end, cost time = %.3f ms, %.3f Mpaths/s, %.3f Mrays/s\n", st.ms_total,
                st.paths / st.ms_total * 1e-3, (st.segments + st.shadow_rays) / st.ms_total * 1e-3);
    if (!dump_path.empty())
    {
        // the pre-clamp sums of the part (rows [A, B) only), for python -m pathtracer_amd.image --merge:
        // "PTXPART1", 12 int32 (width, height, spp_total, row_begin, row_end, sample_begin, sample_end, mode,
        // estimator, depth, 0, 0), uint64 seed, then (row_end - row_begin) * width * 3 doubles, little endian
        FILE *f = std::fopen(dump_path.c_str(), "wb");
        if (!f)
        {
            std::printf("[error] cannot write %s\n", dump_path.c_str());
            return 1;
        }
        const int32_t head[12] = {info.width, info.height, info.spp, rows_a, rows_b, smp_a, smp_b, ro.mode, ro.estimator, info.depth, 0, 0};
        const uint64_t seed = ro.seed;
        const size_t n = static_cast<size_t>(rows_b - rows_a) * info.width * 3;
        bool ok = std::fwrite("PTXPART1", 1, 8, f) == 8 && std::fwrite(head, sizeof(head), 1, f) == 1 && std::fwrite(&seed, 8, 1, f) == 1 &&
                  std::fwrite(sum.data() + static_cast<size_t>(rows_a) * info.width * 3, sizeof(double), n, f) == n;
        ok = (std::fclose(f) == 0) && ok;
        if (!ok)
        {
            std::printf("[error] short write to %s\n", dump_path.c_str());
            return 1;
        }
        std::printf("[debug] rows %d:%d, samples %d:%d of %d -> %s\n", rows_a, rows_b, smp_a, smp_b, info.spp, dump_path.c_str());
        for (ptx_context *c : ctxs)
            ptx_context_destroy(c);
        ptx_scene_destroy(scene);
        return 0;
    }
    ptx_finalize_image(sum.data(), static_cast<int64_t>(info.width) * info.height, img.data());
    if (out_path.empty())
        out_path = ptx_scene_result_path(scene);
    if (ptx_write_ppm(out_path.c_str(), img.data(), info.width, info.height))
        return die("write ppm");
    std::printf("[debug] output img to %s, finished\n", out_path.c_str());
    for (ptx_context *c : ctxs)
        ptx_context_destroy(c);
    ptx_scene_destroy(scene);
    return 0;
}
