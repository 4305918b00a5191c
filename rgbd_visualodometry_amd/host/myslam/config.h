// config.h -- flat "key: value" parameter file, same keys and file format as the reference's
// config/default.yaml (read there through cv::FileStorage: src/config.cpp:25-36, config.h:39-46).
#ifndef MYSLAM_CONFIG_H
#define MYSLAM_CONFIG_H
#include <sstream>

#include "myslam/common_include.h"

namespace myslam {
class Config {
public:
    // Load a parameter file; "%YAML:1.0" header and '#' comments are tolerated.
    // A missing file prints to cerr like the reference and leaves the store empty.
    static void setParameterFile(const std::string& filename);
    // Programmatic override / definition (used by the C wrapper and tests).
    static void set(const std::string& key, const std::string& value);
    static bool has(const std::string& key);
    template <typename T>
    static T get(const std::string& key) {
        T out{};
        std::string raw = raw_value(key);
        std::istringstream ss(raw);
        ss >> out;
        return out;
    }
private:
    static std::string raw_value(const std::string& key);
};
template <>
inline std::string Config::get<std::string>(const std::string& key) { return raw_value(key); }
}  // namespace myslam
#endif
