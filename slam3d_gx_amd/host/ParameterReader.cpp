#include "ParameterReader.h"

#include <cstdlib>
#include <fstream>
#include <iostream>

ParameterReader *g_pParaReader = nullptr;
double camera_fx = 525.0, camera_fy = 525.0, camera_cx = 319.5, camera_cy = 235.5, camera_factor = 1000.0;

static std::string trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r\n\"'");
    if (a == std::string::npos) return "";
    const size_t b = s.find_last_not_of(" \t\r\n\"'");
    return s.substr(a, b - a + 1);
}

ParameterReader::ParameterReader(const std::string &para_file)
{
    std::cout << "init parameterReader, file addr = " << para_file << std::endl;
    std::ifstream fin(para_file.c_str());
    if (!fin) { std::cerr << "cannot open " << para_file << std::endl; return; }
    std::string line;
    while (std::getline(fin, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        if (line.empty() || line[0] == '%') continue;          // "%YAML:1.0" directive
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
        if (!key.empty()) _values[key] = val;
    }
    _ok = true;
    camera_fx = GetDouble("camera_fx", camera_fx); camera_fy = GetDouble("camera_fy", camera_fy);
    camera_cx = GetDouble("camera_cx", camera_cx); camera_cy = GetDouble("camera_cy", camera_cy);
    camera_factor = GetDouble("camera_factor", camera_factor);
    if (GetInt("end_index", 1) < GetInt("start_index", 1))
        std::cerr << "end index should be larger than start index." << std::endl;
}

std::string ParameterReader::GetPara(const std::string &para_name) const
{
    const auto it = _values.find(para_name);
    if (it != _values.end()) return it->second;
    std::cerr << "Unknown parameter: " << para_name << std::endl;
    return "unknown_para_name";
}

double ParameterReader::GetDouble(const std::string &name, double dflt) const
{
    const auto it = _values.find(name);
    return it == _values.end() ? dflt : atof(it->second.c_str());
}

int ParameterReader::GetInt(const std::string &name, int dflt) const
{
    const auto it = _values.find(name);
    return it == _values.end() ? dflt : atoi(it->second.c_str());
}
