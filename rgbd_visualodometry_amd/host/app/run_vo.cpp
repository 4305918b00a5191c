// run_vo.cpp -- the reference's command-line driver (app/run_vo.cpp:27-134) on this framework's host layer:
//   run_vo <parameter_file.yaml>
// reads <dataset_dir>/associate.txt, decodes the PNG pairs, feeds FrontEnd::AddFrame and writes the TUM
// trajectory "timestamp tx ty tz qx qy qz qw" to output_file.  No OpenCV / Boost: PNG over zlib, wall-clock
// timing of AddFrame only (the region the reference times, :104-109).
#include <chrono>
#include <fstream>
#include <iostream>

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
        myslam::FrontEnd::Ptr frontend(new myslam::FrontEnd(myslam::Config::has("device") ? myslam::Config::get<int>("device") : 0, color0.width, color0.height, 1));
        myslam::Backend::Ptr backend;
        if (myslam::Config::get<int>("enable_local_optimization")) {
            std::cout << "Enable local optimization" << std::endl;
            backend = myslam::Backend::Ptr(new myslam::Backend(camera));
            frontend->SetBackend(backend);
        }
        std::cout << "Finish initialization! (compute backend: " << vo_backend_name() << ")\n\n" << std::endl;
        double totalMs = 0; size_t timed = 0;
        for (size_t i = 0; i < entries.size(); ++i) {
            myslam::DecodedImage color, depth;
            if (!myslam::ReadColorBGR(datasetDir + "/" + entries[i].rgbFile, color) || !myslam::ReadDepth16(datasetDir + "/" + entries[i].depthFile, depth)) {
                std::cout << "Frame missing" << std::endl;
                break;
            }
            myslam::Image c, d;
            c.data = color.data.data(); c.rows = color.height; c.cols = color.width; c.stride = 3 * color.width;
            d.data = depth.data.data(); d.rows = depth.height; d.cols = depth.width; d.stride = 2 * depth.width;
            myslam::Frame::Ptr pFrame = myslam::Frame::CreateFrame(entries[i].rgbTime, camera, c, d);
            auto t0 = std::chrono::steady_clock::now();
            frontend->AddFrame(pFrame);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            totalMs += ms; ++timed;
            if (frontend->GetState() == myslam::FrontEnd::LOST) { std::cout << "VO lost" << std::endl; break; }
            char stamp[64];
            snprintf(stamp, sizeof(stamp), "%f", pFrame->timestamp_);          // std::to_string(double) formatting (run_vo.cpp:116)
            myslam::WritePoseLine(fout, stamp, pFrame->GetPose().inverse());
        }
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
