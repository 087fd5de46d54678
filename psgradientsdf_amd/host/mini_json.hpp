// mini_json.hpp — the subset of JSON the voxelPS config files use (one flat object of string / number / bool values,
// config/*.json).  Replaces nlohmann::json in ConfigLoader.h:16-170 (not available in this build environment).
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

namespace psgsdf_host {

struct JsonValue {
    enum Kind { STRING, NUMBER, BOOL, NUL } kind = NUL;
    std::string s; double n = 0; bool b = false;
};

class JsonObject {
    std::map<std::string, JsonValue> kv_;
    std::vector<std::string> order_;
    static void skip(const std::string& t, size_t& i) { while (i < t.size() && std::isspace((unsigned char)t[i])) ++i; }
    static bool parse_string(const std::string& t, size_t& i, std::string& out) {
        if (i >= t.size() || t[i] != '"') return false;
        ++i; out.clear();
        while (i < t.size() && t[i] != '"') {
            if (t[i] == '\\' && i + 1 < t.size()) { char c = t[++i]; out += c == 'n' ? '\n' : c == 't' ? '\t' : c; }
            else out += t[i];
            ++i;
        }
        if (i >= t.size()) return false;
        ++i; return true;
    }
public:
    bool parse(const std::string& text) {
        size_t i = 0; skip(text, i);
        if (i >= text.size() || text[i] != '{') return false;
        ++i;
        while (true) {
            skip(text, i);
            if (i < text.size() && text[i] == '}') return true;
            std::string key; if (!parse_string(text, i, key)) return false;
            skip(text, i); if (i >= text.size() || text[i] != ':') return false; ++i; skip(text, i);
            JsonValue v;
            if (text[i] == '"') { v.kind = JsonValue::STRING; if (!parse_string(text, i, v.s)) return false; }
            else if (!text.compare(i, 4, "true")) { v.kind = JsonValue::BOOL; v.b = true; i += 4; }
            else if (!text.compare(i, 5, "false")) { v.kind = JsonValue::BOOL; v.b = false; i += 5; }
            else if (!text.compare(i, 4, "null")) { v.kind = JsonValue::NUL; i += 4; }
            else { char* end = nullptr; v.n = std::strtod(text.c_str() + i, &end); if (end == text.c_str() + i) return false; v.kind = JsonValue::NUMBER; i = end - text.c_str(); }
            if (!kv_.count(key)) order_.push_back(key);
            kv_[key] = v;
            skip(text, i);
            if (i < text.size() && text[i] == ',') { ++i; continue; }
            if (i < text.size() && text[i] == '}') return true;
            return false;
        }
    }
    bool load(const std::string& path) { std::ifstream f(path); if (!f.is_open()) return false; std::stringstream ss; ss << f.rdbuf(); return parse(ss.str()); }
    bool contains(const std::string& k) const { return kv_.count(k) != 0; }
    std::string str(const std::string& k) const { return kv_.at(k).s; }
    double num(const std::string& k) const { const JsonValue& v = kv_.at(k); return v.kind == JsonValue::BOOL ? (v.b ? 1.0 : 0.0) : v.n; }
    bool boolean(const std::string& k) const { const JsonValue& v = kv_.at(k); return v.kind == JsonValue::BOOL ? v.b : v.n != 0; }
    // `save_conf << std::setw(4) << config` (ConfigLoader.h:161-165): 4-space indented dump, keys sorted like nlohmann's std::map
    void dump(std::ostream& os) const {
        os << "{\n"; size_t k = 0;
        for (auto& kv : kv_) {
            os << "    \"" << kv.first << "\": ";
            const JsonValue& v = kv.second;
            if (v.kind == JsonValue::STRING) os << '"' << v.s << '"'; else if (v.kind == JsonValue::BOOL) os << (v.b ? "true" : "false"); else if (v.kind == JsonValue::NUL) os << "null"; else os << v.n;
            os << (++k < kv_.size() ? ",\n" : "\n");
        }
        os << "}";
    }
};

}  // namespace psgsdf_host
