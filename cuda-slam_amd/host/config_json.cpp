// Configuration reader: a ~150-line recursive-descent JSON reader plus the key/default table of the reference's
// configuration (see configuration.h for the reference lines each rule comes from).
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "configuration.h"

namespace Common {
namespace {

struct Json;
using JsonPtr = std::shared_ptr<Json>;

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonPtr> arr;
    std::map<std::string, JsonPtr> obj;
};

class Reader {
public:
    explicit Reader(const std::string& s) : s_(s) {}
    JsonPtr parse()
    {
        JsonPtr v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s_;
    size_t i_ = 0;

    [[noreturn]] void fail(const char* what) const
    {
        std::ostringstream o;
        o << "JSON: " << what << " at offset " << i_;
        throw std::runtime_error(o.str());
    }
    void ws() { while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) i_++; }
    bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { i_++; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail("unexpected character"); }

    JsonPtr value()
    {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        auto v = std::make_shared<Json>();
        const char c = s_[i_];
        if (c == '{') {
            i_++;
            v->kind = Json::Object;
            if (eat('}')) return v;
            do {
                ws();
                const std::string key = string();
                expect(':');
                v->obj[key] = value();
            } while (eat(','));
            expect('}');
        } else if (c == '[') {
            i_++;
            v->kind = Json::Array;
            if (eat(']')) return v;
            do v->arr.push_back(value()); while (eat(','));
            expect(']');
        } else if (c == '"') {
            v->kind = Json::String;
            v->str = string();
        } else if (s_.compare(i_, 4, "true") == 0) { v->kind = Json::Bool; v->b = true; i_ += 4; }
        else if (s_.compare(i_, 5, "false") == 0) { v->kind = Json::Bool; v->b = false; i_ += 5; }
        else if (s_.compare(i_, 4, "null") == 0) { i_ += 4; }
        else {
            char* end = nullptr;
            v->kind = Json::Number;
            v->num = std::strtod(s_.c_str() + i_, &end);
            if (end == s_.c_str() + i_) fail("bad value");
            i_ = (size_t)(end - s_.c_str());
        }
        return v;
    }

    std::string string()
    {
        if (i_ >= s_.size() || s_[i_] != '"') fail("expected string");
        i_++;
        std::string out;
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\' && i_ < s_.size()) {
                const char e = s_[i_++];
                switch (e) {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                default: c = e; break;   // \" \\ \/ ; \uXXXX is not needed by any config key or path we read
                }
            }
            out.push_back(c);
        }
        if (i_ >= s_.size()) fail("unterminated string");
        i_++;
        return out;
    }
};

const Json* find(const Json& root, const char* key)
{
    auto it = root.obj.find(key);
    return it == root.obj.end() ? nullptr : it->second.get();
}

template <typename T>
std::optional<T> number(const Json& root, const char* key)
{
    const Json* j = find(root, key);
    if (!j) return std::nullopt;
    if (j->kind != Json::Number) throw std::runtime_error(std::string("key '") + key + "' must be a number");
    return static_cast<T>(j->num);
}

template <typename T>
T number_or(const Json& root, const char* key, T dflt)
{
    auto v = number<T>(root, key);
    return v ? *v : dflt;
}

bool bool_or(const Json& root, const char* key, bool dflt)
{
    const Json* j = find(root, key);
    if (!j) return dflt;
    if (j->kind != Json::Bool) throw std::runtime_error(std::string("key '") + key + "' must be a boolean");
    return j->b;
}

std::optional<std::string> text(const Json& root, const char* key)
{
    const Json* j = find(root, key);
    if (!j) return std::nullopt;
    if (j->kind != Json::String) throw std::runtime_error(std::string("key '") + key + "' must be a string");
    return j->str;
}

constexpr const char* kDefaultPath = "config/default.json";

bool file_exists(const std::string& p)
{
    std::ifstream f(p);
    return f.good();
}

bool load_file(const std::string& path, Configuration* out)
{
    std::ifstream stream(path);
    std::stringstream ss;
    ss << stream.rdbuf();
    return ParseConfiguration(ss.str(), out);
}

}  // namespace

bool ParseConfiguration(const std::string& json_text, Configuration* out)
{
    Configuration cfg;
    bool correct = true;
    try {
        Reader rd(json_text);
        const JsonPtr rootp = rd.parse();
        const Json& root = *rootp;
        if (root.kind != Json::Object) throw std::runtime_error("top level must be an object");

        // "method": required (configparser.cpp:70-92)
        if (auto m = text(root, "method")) {
            if (*m == "icp") cfg.ComputationMethod_ = ComputationMethod::Icp;
            else if (*m == "nicp") cfg.ComputationMethod_ = ComputationMethod::NoniterativeIcp;
            else if (*m == "cpd") cfg.ComputationMethod_ = ComputationMethod::Cpd;
            else { printf("Parsing error: Computational method %s not supported\n", m->c_str()); correct = false; }
        } else { printf("Parsing error: Required parameter method not found\n"); correct = false; }

        // cloud paths: required (:94-103)
        auto bp = text(root, "before-path"), ap = text(root, "after-path");
        if (bp && ap) { cfg.BeforePath = *bp; cfg.AfterPath = *ap; }
        else { printf("Parsing error: Required parameter before-path / after-path not found\n"); correct = false; }

        // "policy" (:105-126)
        if (auto p = text(root, "policy")) {
            if (*p == "parallel") cfg.ExecutionPolicy_ = ExecutionPolicy::Parallel;
            else if (*p == "sequential") cfg.ExecutionPolicy_ = ExecutionPolicy::Sequential;
            else { printf("Parsing warning: Execution policy %s not supported\n", p->c_str()); correct = false; }
        }

        // explicit transformation (:128-162)
        const Json* tr = find(root, "translation");
        const Json* ro = find(root, "rotation");
        const float scale = number_or<float>(root, "scale", 1.0f);
        if (tr && ro) {
            if (tr->kind != Json::Array || ro->kind != Json::Array || tr->arr.size() != 3 || ro->arr.size() != 9) {
                printf("Parsing error: Wrong translation or rotation size\n");
                correct = false;
            } else {
                Mat3 R;
                for (int x = 0; x < 3; x++)
                    for (int y = 0; y < 3; y++) {
                        const Json& e = *ro->arr[(size_t)(x * 3 + y)];
                        if (e.kind != Json::Number) throw std::runtime_error("rotation entries must be numbers");
                        R[y][x] = (float)e.num;      // file is row-major, Mat3 is m[col][row]
                    }
                Vec3 t;
                for (int i = 0; i < 3; i++) {
                    const Json& e = *tr->arr[(size_t)i];
                    if (e.kind != Json::Number) throw std::runtime_error("translation entries must be numbers");
                    t[i] = (float)e.num;
                }
                cfg.Transformation = std::make_pair(scale * R, t);
            }
        }

        // random transformation parameters (:164-187)
        auto trr = number<float>(root, "translation-range"), ror = number<float>(root, "rotation-range");
        if (trr && ror) cfg.TransformationParameters = std::make_pair(*ror, *trr);

        // the rest (:189-257)
        cfg.MaxIterations = number<int>(root, "max-iterations");
        cfg.CloudBeforeResize = number<int>(root, "cloud-before-resize");
        cfg.CloudAfterResize = number<int>(root, "cloud-after-resize");
        cfg.CloudSpread = number<float>(root, "cloud-spread");
        cfg.RandomSeed = number<int>(root, "random-seed");
        cfg.NoiseAffectedPointsBefore = number<float>(root, "noise-affected-points-before");
        cfg.NoiseAffectedPointsAfter = number<float>(root, "noise-affected-points-after");
        cfg.ShowVisualisation = bool_or(root, "show-visualisation", false);
        cfg.MaxDistanceSquared = number_or<float>(root, "max-distance-squared", 1000.f);
        cfg.ApproximationType_ = ApproximationType::Hybrid;
        if (auto a = text(root, "approximation-type")) {
            if (*a == "full") cfg.ApproximationType_ = ApproximationType::Full;
            else if (*a == "none") cfg.ApproximationType_ = ApproximationType::None;
            else cfg.ApproximationType_ = ApproximationType::Hybrid;   // unknown strings fall back to hybrid (:226-229)
        }
        cfg.NicpBatchSize = number_or<int>(root, "nicp-batch-size", 16);
        cfg.NicpIterations = number_or<int>(root, "nicp-iterations", 32);
        cfg.NicpSubcloudSize = number_or<int>(root, "nicp-subcloud-size", 1000);
        cfg.CpdWeight = number_or<float>(root, "cpd-weight", 0.3f);
        cfg.CpdConstScale = bool_or(root, "cpd-const-scale", false);
        cfg.CpdTolerance = number_or<float>(root, "cpd-tolerance", 1e-3f);
        cfg.ConvergenceEpsilon = number_or<float>(root, "convergence-epsilon", 1e-3f);
        cfg.NoiseIntensityBefore = number_or<float>(root, "noise-intensity-before", 0.1f);
        cfg.NoiseIntensityAfter = number_or<float>(root, "noise-intensity-after", 0.1f);
        cfg.AdditionalOutliersBefore = number_or<int>(root, "additional-outliers-before", 0);
        cfg.AdditionalOutliersAfter = number_or<int>(root, "additional-outliers-after", 0);
        cfg.RatioOfFarField = number_or<float>(root, "fgt-ratio-of-far-field", 10.0f);
        cfg.OrderOfTruncation = number_or<int>(root, "fgt-order-of-truncation", 8);

        // validation (:259-266)
        if (!cfg.Transformation && !cfg.TransformationParameters) {
            printf("Parsing error: transformation or transformation parameters have to be provided\n");
            correct = false;
        }
    } catch (const std::exception& ex) {
        printf("Parsing error: %s\n", ex.what());
        correct = false;
    }
    if (correct && out) *out = cfg;
    return correct;
}

bool LoadConfigurationFromArgs(int argc, char** argv, Configuration* out)
{
    if (argc == 1) {
        printf("No config passed, loading: %s\n", kDefaultPath);
        return load_file(kDefaultPath, out);
    }
    if (argc == 2) {
        const std::string path = argv[1];
        if (file_exists(path)) {
            printf("Loading config from: %s\n", path.c_str());
            return load_file(path, out);
        }
        printf("File: %s does not exist, loading default config\n", path.c_str());
        return load_file(kDefaultPath, out);
    }
    printf("Usage: %s (config_path)\n", argv[0]);
    printf("Loading default config\n");
    return load_file(kDefaultPath, out);
}

void Configuration::Print() const
{
    static const char* methods[] = {"icp", "nicp", "cpd"};
    static const char* approx[] = {"none", "full", "hybrid"};
    printf("===============================\n");
    printf("Method: %s\n", methods[(int)ComputationMethod_]);
    printf("Before path: %s\nAfter path: %s\n", BeforePath.c_str(), AfterPath.c_str());
    if (MaxIterations) printf("Max iterations: %d\n", *MaxIterations);
    if (CloudSpread) printf("Cloud spread: %f\n", *CloudSpread);
    if (RandomSeed) printf("Random seed: %d\n", *RandomSeed);
    printf("Max distance squared: %f\nConvergence epsilon: %f\n", MaxDistanceSquared, ConvergenceEpsilon);
    printf("Approximation type: %s\nCpd weight: %f, const scale: %d, tolerance: %f\n", approx[(int)ApproximationType_], CpdWeight,
           (int)CpdConstScale, CpdTolerance);
    printf("===============================\n");
}

}  // namespace Common
