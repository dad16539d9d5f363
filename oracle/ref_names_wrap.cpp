// TEST INFRASTRUCTURE (oracle/): C wrapper around the REFERENCE's own checkpoint-name rules — convert_tensor_name, /root/reference/src/name_conversion.cpp:1346,
// compiled from where it lies (oracle/Makefile; oracle/stubs/ stands in for the absent ggml headers model.h mentions).  Used by tests/golden/
// make_names_golden.py (committed golden table) and, when present, live by tests/test_name_conversion.py.  Never loaded by the product.
#include <cstring>
#include <string>
#include <vector>

#include "name_conversion.h"

// the string helpers of src/core/util.cpp (:34-54, 398-413) name_conversion.cpp links against, restated (util.cpp itself needs ggml): suffix / prefix /
// substring tests and a split on one character
bool ends_with(const std::string& str, const std::string& ending) { return str.length() >= ending.length() && str.compare(str.length() - ending.length(), ending.length(), ending) == 0; }
bool starts_with(const std::string& str, const std::string& start) { return str.find(start) == 0; }
bool contains(const std::string& str, const std::string& substr) { return str.find(substr) != std::string::npos; }
std::vector<std::string> split_string(const std::string& str, char delimiter) {  // util.cpp:398-413
    std::vector<std::string> result;
    size_t start = 0, end = str.find(delimiter);
    while (end != std::string::npos) {
        result.push_back(str.substr(start, end - start));
        start = end + 1;
        end   = str.find(delimiter, start);
    }
    result.push_back(str.substr(start));
    return result;
}
size_t ggml_type_size(enum ggml_type) { return 4; }
int64_t ggml_blck_size(enum ggml_type) { return 1; }
const char* ggml_type_name(enum ggml_type) { return "stub"; }

// family: 0 = SD1.x, 1 = SDXL, 2 = SD3.x, 3 = FLUX.1
extern "C" __attribute__((visibility("default"))) int ref_convert_tensor_name(const char* name, int family, char* out, int cap) {
    const SDVersion v   = family == 0 ? VERSION_SD1 : (family == 1 ? VERSION_SDXL : (family == 2 ? VERSION_SD3 : VERSION_FLUX));
    const std::string r = convert_tensor_name(std::string(name), v);
    if ((int)r.size() + 1 > cap) return -1;
    std::memcpy(out, r.c_str(), r.size() + 1);
    return (int)r.size();
}
