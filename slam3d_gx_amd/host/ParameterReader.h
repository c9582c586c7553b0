// ParameterReader.h -- reads the reference's flag file (parameters.yaml) with the reference's accessor
// shape: GetPara(name) returns the value as a *string*, unknown keys give "unknown_para_name" plus a
// message on stderr (src/ParameterReader.cpp:69-123).  The reference parses it with yaml-cpp 0.3, which
// this image lacks; the file is a flat `key: value` list (parameters.yaml:1-98), so a line parser is
// enough.  New ICP keys fall back to defaults when absent.
#pragma once
#include <map>
#include <string>

class ParameterReader {
 public:
    explicit ParameterReader(const std::string &para_file = "./parameters.yaml");
    // same contract as the reference's GetPara: string out, "unknown_para_name" when missing
    std::string GetPara(const std::string &para_name) const;
    bool Has(const std::string &para_name) const { return _values.count(para_name) != 0; }
    double GetDouble(const std::string &name, double dflt) const;
    int GetInt(const std::string &name, int dflt) const;
    bool ok() const { return _ok; }

 private:
    std::map<std::string, std::string> _values;
    bool _ok = false;
};

// the reference keeps these as process globals set by the reader's constructor (src/ParameterReader.cpp:9,55-59)
extern ParameterReader *g_pParaReader;
extern double camera_fx, camera_fy, camera_cx, camera_cy, camera_factor;
