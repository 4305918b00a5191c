// dataset.h -- TUM RGB-D dataset I/O for the run_vo driver: associate.txt parser
// (reference app/run_vo.cpp:47-63), PNG decode replacing cv::imread (:91-92: 8-bit colour -> BGR 8UC3,
// 16-bit grey depth -> 16UC1) over zlib, and the trajectory line writer (:19-25).
#ifndef MYSLAM_DATASET_H
#define MYSLAM_DATASET_H
#include <fstream>

#include "myslam/common_include.h"

namespace myslam {

struct AssociateEntry { double rgbTime, depthTime; std::string rgbTimeText, rgbFile, depthFile; };
// lines "rgbT rgbFile depthT depthFile"; stops at the first short line like the reference's loop
std::vector<AssociateEntry> ReadAssociateFile(const std::string& path);

struct DecodedImage { int width = 0, height = 0, channels = 0, bitDepth = 0; std::vector<uint8_t> data; };   // row-major, tight
// Non-interlaced PNG: colour types 0 (grey 8/16), 2 (RGB 8), 6 (RGBA 8).  16-bit samples are returned in
// host byte order.  Returns false (and leaves `out` empty) on any malformed / unsupported file.
bool DecodePng(const std::string& path, DecodedImage& out);
// cv::imread(path) equivalent: BGR 8UC3 (grey is replicated, alpha dropped)
bool ReadColorBGR(const std::string& path, DecodedImage& out);
// cv::imread(path, -1) equivalent for TUM depth: 16UC1
bool ReadDepth16(const std::string& path, DecodedImage& out);

// "timestamp tx ty tz qx qy qz qw" with the camera-to-world pose (run_vo.cpp:19-25)
void WritePoseLine(std::ostream& os, const std::string& stamp, const SE3& Twc);

}  // namespace myslam
#endif
