// config.cpp -- see myslam/config.h (reference src/config.cpp:25-45).
#include "myslam/config.h"

#include <fstream>

namespace myslam {
namespace {
std::mutex g_mu;
std::map<std::string, std::string>& store() { static std::map<std::string, std::string> s; return s; }
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n\"'"), b = s.find_last_not_of(" \t\r\n\"'");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
}  // namespace

void Config::setParameterFile(const std::string& filename) {
    std::ifstream fin(filename);
    if (!fin) { std::cerr << "parameter file " << filename << " does not exist." << std::endl; return; }
    std::unique_lock<std::mutex> lk(g_mu);
    std::string line;
    while (std::getline(fin, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        if (line.empty() || line[0] == '%') continue;          // "%YAML:1.0"
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
        if (!key.empty()) store()[key] = val;
    }
}
void Config::set(const std::string& key, const std::string& value) { std::unique_lock<std::mutex> lk(g_mu); store()[key] = value; }
bool Config::has(const std::string& key) { std::unique_lock<std::mutex> lk(g_mu); return store().count(key) != 0; }
std::string Config::raw_value(const std::string& key) {
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = store().find(key);
    return it == store().end() ? std::string() : it->second;
}
}  // namespace myslam
