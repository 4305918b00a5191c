// run_vo.cpp -- the reference's command-line driver (app/run_vo.cpp:27-134) on this framework's host layer:
//   run_vo <parameter_file.yaml>
// reads <dataset_dir>/associate.txt, decodes the PNG pairs, feeds FrontEnd::AddFrame and writes the TUM
// trajectory "timestamp tx ty tz qx qy qz qw" to output_file.  No OpenCV / Boost: PNG over zlib, wall-clock
// timing of AddFrame only (the region the reference times, :104-109).
// Optional keys (beyond the reference's default.yaml): lookahead_frames (default 1) decodes that many frame pairs ahead on
// decode_threads host threads -- the next chunk is decoded while the current one is tracked -- and hands each chunk to
// FrontEnd::PrefetchFrames (batched ORB); track_batch and backend_lag_frames are read by FrontEnd / Backend.  The
// trajectory does not depend on any of them.
#include <chrono>
#include <fstream>
#include <future>
#include <iostream>
#include <thread>

#include "myslam/backend.h"
#include "myslam/config.h"
#include "myslam/dataset.h"
#include "myslam/frontend.h"
#include "myslam/mapmanager.h"

int main(int argc, char** argv) {
    if (argc != 2) { std::cout << "usage: run_vo parameter_file" << std::endl; return 1; }
    myslam::Config::setParameterFile(argv[1]);
    const std::string datasetDir = myslam::Config::get<std::string>("dataset_dir");
    const std::string entry = datasetDir + "/associate.txt";
    std::cout << "Path of dataset: " << entry << std::endl;
    auto entries = myslam::ReadAssociateFile(entry);
    if (entries.empty()) { std::cout << "please generate the associate file called associate.txt!" << std::endl; return 1; }
    std::cout << "Total " << entries.size() << " images from dataset\n\n";

    const std::string outputPath = myslam::Config::get<std::string>("output_file");
    std::ofstream fout(outputPath);
    fout << "# estimated trajectory format" << std::endl;
    fout << "# timestamp tx ty tz qx qy qz qw" << std::endl;

    std::cout << "Initializing VO system ..." << std::endl;
    int rc = 0;
    try {
        myslam::DecodedImage color0, depth0;
        if (!myslam::ReadColorBGR(datasetDir + "/" + entries[0].rgbFile, color0)) { std::cout << "Frame missing" << std::endl; return 1; }
        myslam::Camera::Ptr camera(new myslam::Camera);
        const int lookahead = std::max(1, myslam::Config::has("lookahead_frames") ? myslam::Config::get<int>("lookahead_frames") : 1);
        const int decodeThreads = std::max(1, myslam::Config::has("decode_threads") ? myslam::Config::get<int>("decode_threads")
                                                                                    : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
        myslam::FrontEnd::Ptr frontend(new myslam::FrontEnd(myslam::Config::has("device") ? myslam::Config::get<int>("device") : 0, color0.width, color0.height, lookahead));
        myslam::Backend::Ptr backend;
        if (myslam::Config::get<int>("enable_local_optimization")) {
            std::cout << "Enable local optimization" << std::endl;
            backend = myslam::Backend::Ptr(new myslam::Backend(camera));
            frontend->SetBackend(backend);
        }
        // Several ranks of one job (one per GPU) on the SAME dataset: world_size / rank / rccl_id_file in the parameter file.  Every rank tracks every frame; the RANSAC
        // hypotheses of a frame (shard_hypotheses, default on) and / or the local BA's edges (shard_ba, default off) are shared out and exchanged by RCCL all-reduces
        // on the launch chains' streams.  Every rank writes the same trajectory: give each its own output_file.
        const int world = myslam::Config::has("world_size") ? myslam::Config::get<int>("world_size") : 1;
        if (world > 1) {
            const int rank = myslam::Config::get<int>("rank");
            const std::string idFile = myslam::Config::get<std::string>("rccl_id_file");
            if (!myslam::Config::has("shard_hypotheses") || myslam::Config::get<int>("shard_hypotheses")) frontend->ShardHypothesesOverRanks(rank, world, idFile);
            if (backend && myslam::Config::has("shard_ba") && myslam::Config::get<int>("shard_ba")) backend->ShardOverRanks(rank, world, idFile + ".ba");
            std::cout << "rank " << rank << " of " << world << " (RCCL exchanges on the launch chains' streams)" << std::endl;
        }
        std::cout << "Finish initialization! (compute backend: " << vo_backend_name() << ", lookahead " << lookahead << ")\n\n" << std::endl;

        // one chunk = up to `lookahead` decoded frame pairs; ok[i] false marks the first missing / undecodable frame
        struct Chunk { size_t first = 0; std::vector<myslam::DecodedImage> color, depth; std::vector<char> ok; };
        auto decodeChunk = [&](size_t first) {
            Chunk ch;
            ch.first = first;
            const size_t n = std::min<size_t>(lookahead, entries.size() - first);
            ch.color.resize(n); ch.depth.resize(n); ch.ok.assign(n, 0);
            auto work = [&](size_t t0) {
                for (size_t j = t0; j < n; j += (size_t)decodeThreads)
                    ch.ok[j] = myslam::ReadColorBGR(datasetDir + "/" + entries[first + j].rgbFile, ch.color[j]) &&
                               myslam::ReadDepth16(datasetDir + "/" + entries[first + j].depthFile, ch.depth[j]);
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < decodeThreads && (size_t)t < n; ++t) pool.emplace_back(work, (size_t)t);
            work(0);
            for (auto& th : pool) th.join();
            return ch;
        };

        double totalMs = 0; size_t timed = 0;
        bool stop = false;
        std::future<Chunk> next = std::async(std::launch::async, decodeChunk, (size_t)0);
        for (size_t first = 0; first < entries.size() && !stop; first += (size_t)lookahead) {
            Chunk ch = next.get();
            if (first + (size_t)lookahead < entries.size()) next = std::async(std::launch::async, decodeChunk, first + (size_t)lookahead);
            std::vector<myslam::Frame::Ptr> frames;
            for (size_t j = 0; j < ch.ok.size(); ++j) {
                if (!ch.ok[j]) { std::cout << "Frame missing" << std::endl; stop = true; break; }
                myslam::Image c, d;
                c.data = ch.color[j].data.data(); c.rows = ch.color[j].height; c.cols = ch.color[j].width; c.stride = 3 * ch.color[j].width;
                d.data = ch.depth[j].data.data(); d.rows = ch.depth[j].height; d.cols = ch.depth[j].width; d.stride = 2 * ch.depth[j].width;
                frames.push_back(myslam::Frame::CreateFrame(entries[first + j].rgbTime, camera, c, d));
            }
            auto t0 = std::chrono::steady_clock::now();
            if (lookahead > 1 && !frames.empty()) frontend->PrefetchFrames(frames);        // one batched ORB launch chain for the chunk
            totalMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            for (auto& pFrame : frames) {
                t0 = std::chrono::steady_clock::now();
                frontend->AddFrame(pFrame);
                totalMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                ++timed;
                if (frontend->GetState() == myslam::FrontEnd::LOST) { std::cout << "VO lost" << std::endl; stop = true; break; }
                char stamp[64];
                snprintf(stamp, sizeof(stamp), "%f", pFrame->timestamp_);          // std::to_string(double) formatting (run_vo.cpp:116)
                myslam::WritePoseLine(fout, stamp, pFrame->GetPose().inverse());
            }
        }
        if (next.valid()) next.wait();
        if (backend) backend->Stop();
        if (timed) std::cout << "Frames: " << timed << ", mean AddFrame time (ms): " << totalMs / timed << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "run_vo: " << e.what() << std::endl;
        rc = 2;
    }
    fout.close();
    std::cout << "Finished. \nWrote trajectory to " << outputPath << std::endl;
    return rc;
}
