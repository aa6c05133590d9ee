// onnx_reader.cpp -- minimal ONNX (protobuf) reader for the two model files of the reference
// CLI: `speakerDiarizer segment.onnx embedding.onnx file.wav` (README.md:39, sd.cpp:3428-3430).
// Replaces Ort::Session's model parsing (onnx_model.cc:73-104) for exactly the two graphs the
// reference's exporters produce with torch.onnx.export, opset 17:
//   segment/export2.py:42-52     input "signal"[B,1,T]           -> "segments"[B,293,3]
//   embeddings/export3.py:177-189 inputs "feats"[B,T,201,2], "wav_lens"[B] -> "embedding"[B,1,192]
// No ONNX runtime is involved: the graph is only walked to pull the weights out, in execution
// order, into the library's weight pack (the kernels implement the architectures directly):
//   PyanNet : InstanceNormalization x4, Conv x3, LSTM x4 (W/R/B, gate order iofc), MatMul+Add x3
//   ECAPA   : MatMul with the constant [201,80] mel matrix, Conv x38, BatchNormalization x31
// Anything else (unexpected op counts, non-constant weights, external data) is reported as
// SD_ERR_MODEL with the reason; nothing is guessed.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

struct PB {                      // protobuf wire reader over a byte range
    const uint8_t* p; const uint8_t* e; bool ok = true;
    PB(const uint8_t* b, size_t n) : p(b), e(b + n) {}
    bool done() const { return p >= e || !ok; }
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (p < e) { const uint8_t c = *p++; v |= (uint64_t)(c & 0x7f) << sh; if (!(c & 0x80)) return v; sh += 7; if (sh > 63) break; }
        ok = false; return 0;
    }
    bool tag(int& field, int& wt) { if (done()) return false; const uint64_t t = varint(); field = (int)(t >> 3); wt = (int)(t & 7); return ok; }
    PB sub() { const uint64_t n = varint(); if (!ok || (uint64_t)(e - p) < n) { ok = false; return PB(p, 0); } PB s(p, (size_t)n); p += n; return s; }
    void skip(int wt) {
        if (wt == 0) (void)varint();
        else if (wt == 1) { if (e - p < 8) ok = false; else p += 8; }
        else if (wt == 2) (void)sub();
        else if (wt == 5) { if (e - p < 4) ok = false; else p += 4; }
        else ok = false;
    }
    std::string str() { PB s = sub(); return std::string((const char*)s.p, (size_t)(s.e - s.p)); }
};

struct OTensor { std::vector<int64_t> dims; int dtype = 0; std::vector<float> f; std::vector<int64_t> i64; bool external = false; std::string name; };
struct OAttr { std::string name; int64_t i = 0; float f = 0; std::string s; std::vector<int64_t> ints; OTensor t; bool has_t = false; };
struct ONode { std::string op, name; std::vector<std::string> in, out; std::vector<OAttr> attrs;
               const OAttr* attr(const char* n) const { for (auto& a : attrs) if (a.name == n) return &a; return nullptr; } };
struct OGraph { std::vector<ONode> nodes; std::map<std::string, OTensor> init; };

bool parse_tensor(PB pb, OTensor& t)
{
    std::string raw;
    int f, wt;
    while (pb.tag(f, wt)) {
        if (f == 1) { if (wt == 2) { PB s = pb.sub(); while (!s.done()) t.dims.push_back((int64_t)s.varint()); } else t.dims.push_back((int64_t)pb.varint()); }
        else if (f == 2 && wt == 0) t.dtype = (int)pb.varint();
        else if (f == 4) { if (wt == 2) { PB s = pb.sub(); while (s.e - s.p >= 4) { float v; memcpy(&v, s.p, 4); s.p += 4; t.f.push_back(v); } } else if (wt == 5) { if (pb.e - pb.p < 4) { pb.ok = false; break; } float v; memcpy(&v, pb.p, 4); pb.p += 4; t.f.push_back(v); } else pb.skip(wt); }
        else if (f == 7) { if (wt == 2) { PB s = pb.sub(); while (!s.done()) t.i64.push_back((int64_t)s.varint()); } else t.i64.push_back((int64_t)pb.varint()); }
        else if (f == 8 && wt == 2) t.name = pb.str();
        else if (f == 9 && wt == 2) raw = pb.str();
        else if (f == 10 && wt == 2) { PB s = pb.sub(); while (s.e - s.p >= 8) { double v; memcpy(&v, s.p, 8); s.p += 8; t.f.push_back((float)v); } }
        else if (f == 13 || f == 14) { t.external = true; pb.skip(wt); }
        else pb.skip(wt);
    }
    if (!pb.ok) return false;
    if (!raw.empty()) {
        if (t.dtype == 1) { t.f.resize(raw.size() / 4); if (!t.f.empty()) memcpy(t.f.data(), raw.data(), t.f.size() * 4); }
        else if (t.dtype == 11) { const size_t n = raw.size() / 8; t.f.resize(n); for (size_t k = 0; k < n; ++k) { double v; memcpy(&v, raw.data() + 8 * k, 8); t.f[k] = (float)v; } }
        else if (t.dtype == 7) { t.i64.resize(raw.size() / 8); if (!t.i64.empty()) memcpy(t.i64.data(), raw.data(), t.i64.size() * 8); }
        else if (t.dtype == 9) { t.i64.resize(raw.size()); for (size_t k = 0; k < raw.size(); ++k) t.i64[k] = raw[k] != 0; }
        else if (t.dtype == 6) { const size_t n = raw.size() / 4; t.i64.resize(n); for (size_t k = 0; k < n; ++k) { int32_t v; memcpy(&v, raw.data() + 4 * k, 4); t.i64[k] = v; } }
    }
    return true;
}

bool parse_attr(PB pb, OAttr& a)
{
    int f, wt;
    while (pb.tag(f, wt)) {
        if (f == 1 && wt == 2) a.name = pb.str();
        else if (f == 2 && wt == 5) { if (pb.e - pb.p < 4) { pb.ok = false; break; } memcpy(&a.f, pb.p, 4); pb.p += 4; }
        else if (f == 3 && wt == 0) a.i = (int64_t)pb.varint();
        else if (f == 4 && wt == 2) a.s = pb.str();
        else if (f == 5 && wt == 2) { a.has_t = true; if (!parse_tensor(pb.sub(), a.t)) return false; }
        else if (f == 8) { if (wt == 2) { PB s = pb.sub(); while (!s.done()) a.ints.push_back((int64_t)s.varint()); } else a.ints.push_back((int64_t)pb.varint()); }
        else pb.skip(wt);
    }
    return pb.ok;
}

bool parse_node(PB pb, ONode& n)
{
    int f, wt;
    while (pb.tag(f, wt)) {
        if (f == 1 && wt == 2) n.in.push_back(pb.str());
        else if (f == 2 && wt == 2) n.out.push_back(pb.str());
        else if (f == 3 && wt == 2) n.name = pb.str();
        else if (f == 4 && wt == 2) n.op = pb.str();
        else if (f == 5 && wt == 2) { OAttr a; if (!parse_attr(pb.sub(), a)) return false; n.attrs.push_back(std::move(a)); }
        else pb.skip(wt);
    }
    return pb.ok;
}

bool parse_graph(PB pb, OGraph& g)
{
    int f, wt;
    while (pb.tag(f, wt)) {
        if (f == 1 && wt == 2) { ONode n; if (!parse_node(pb.sub(), n)) return false; g.nodes.push_back(std::move(n)); }
        else if (f == 5 && wt == 2) { OTensor t; if (!parse_tensor(pb.sub(), t)) return false; g.init[t.name] = std::move(t); }
        else pb.skip(wt);
    }
    // Constant nodes act as initializers under their output name
    for (auto& n : g.nodes)
        if (n.op == "Constant" && !n.out.empty()) {
            const OAttr* a = n.attr("value");
            OTensor t; bool have = false;
            if (a && a->has_t) { t = a->t; have = true; }
            else if ((a = n.attr("value_float"))) { t.dtype = 1; t.f = {a->f}; have = true; }
            else if ((a = n.attr("value_int"))) { t.dtype = 7; t.i64 = {a->i}; have = true; }
            else if ((a = n.attr("value_ints"))) { t.dtype = 7; t.i64 = a->ints; t.dims = {(int64_t)a->ints.size()}; have = true; }
            if (have) { t.name = n.out[0]; g.init[n.out[0]] = std::move(t); }
        }
    return pb.ok;
}

// ---- constant evaluation of the weight-shuffling ops an exporter leaves behind when it does not fold them
// (torch.onnx.export with do_constant_folding=False builds the LSTM W / R / B by Slice + Concat + Unsqueeze of the PyTorch
// parameters and the Linear weights by Transpose): a node whose inputs are all constants becomes a constant itself.
// Only data movement is evaluated -- no arithmetic -- so the extracted weights stay the exporter's bits.
// which of the two value fields a tensor's elements live in is decided by where the data IS (a malformed file may announce one element type and
// fill the other field); every accessor below goes through this one rule
bool is_int(const OTensor& t) { return !t.i64.empty() || (t.f.empty() && (t.dtype == 7 || t.dtype == 6 || t.dtype == 9)); }
int64_t numel(const OTensor& t) { int64_t n = 1; for (auto d : t.dims) n *= d; return n; }
// a tensor the constant folder may touch: non-negative dimensions, a bounded element count (no overflow, no giant resize) and exactly
// that many stored elements -- a malformed model is skipped here and reported by the layer extraction ("... is not a constant")
bool sane(const OTensor& t)
{
    int64_t n = 1;
    for (auto d : t.dims) {
        if (d < 0 || d > (int64_t)1 << 31) return false;
        if (d != 0 && n > ((int64_t)1 << 31) / d) return false;
        n *= d;
    }
    // the storage must be the one the element type says (an int tensor whose data sits in the float field would be indexed through the empty one)
    if (is_int(t)) return t.f.empty() && (int64_t)t.i64.size() == n;
    return t.i64.empty() && (int64_t)t.f.size() == n;
}
// copy a strided N-d view: out[idx] = in[sum (start_a + idx_a * step_a) * stride_a]
void gather_nd(const OTensor& in, const std::vector<int64_t>& odims, const std::vector<int64_t>& start, const std::vector<int64_t>& step,
               const std::vector<int64_t>& istride, OTensor& out)
{
    out.dims = odims; out.dtype = in.dtype;
    const int64_t n = numel(out);
    const bool ii = is_int(in);
    if (ii) out.i64.resize((size_t)n); else out.f.resize((size_t)n);
    std::vector<int64_t> idx(odims.size(), 0);
    for (int64_t k = 0; k < n; ++k) {
        int64_t src = 0;
        for (size_t a = 0; a < odims.size(); ++a) src += (start[a] + idx[a] * step[a]) * istride[a];
        if (ii) out.i64[(size_t)k] = in.i64[(size_t)src]; else out.f[(size_t)k] = in.f[(size_t)src];
        for (int a = (int)odims.size() - 1; a >= 0; --a) { if (++idx[(size_t)a] < odims[(size_t)a]) break; idx[(size_t)a] = 0; }
    }
}
std::vector<int64_t> strides_of(const std::vector<int64_t>& dims)
{
    std::vector<int64_t> s(dims.size(), 1);
    for (int a = (int)dims.size() - 2; a >= 0; --a) s[(size_t)a] = s[(size_t)a + 1] * dims[(size_t)a + 1];
    return s;
}
// ---- arithmetic on constants.  The TorchScript exporter folds only a short list of operators; what it leaves behind in a graph whose
// first convolution is a PARAMETRISED filter bank -- pyannote's SincNet front end computes its 80 x 251 filters from 2 x 40 learnable
// frequencies with Abs, Add, Clip, MatMul, Sin, Cos, Sub, Div, Mul, a flip (Slice with step -1), Concat and Reshape (asteroid
// ParamSincFB.filters()) -- is a constant sub-graph in front of the Conv's weight input.  It is evaluated here in float32 / int64 the way
// an ONNX runtime would evaluate it (element-wise with numpy broadcasting); only nodes ALL of whose inputs are constants are touched,
// so nothing that depends on the audio is ever computed.  (sin / cos are the host libm's: within an ulp of another runtime's.)
bool is_floaty(const OTensor& t) { return !is_int(t); }
double getv(const OTensor& t, int64_t k) { return is_floaty(t) ? (double)t.f[(size_t)k] : (double)t.i64[(size_t)k]; }
bool broadcast_dims(const std::vector<const OTensor*>& in, std::vector<int64_t>& od)
{
    size_t R = 0;
    for (auto* t : in) R = std::max(R, t->dims.size());
    od.assign(R, 1);
    for (auto* t : in)
        for (size_t a = 0; a < t->dims.size(); ++a) {
            const size_t o = R - t->dims.size() + a;
            const int64_t d = t->dims[a];
            if (d == od[o] || d == 1) continue;
            if (od[o] == 1) od[o] = d; else return false;
        }
    return true;
}
// offset of output element `idx` (multi-index over od) inside t under broadcasting
int64_t bc_offset(const OTensor& t, const std::vector<int64_t>& od, const std::vector<int64_t>& idx)
{
    const size_t R = od.size(), r = t.dims.size();
    int64_t off = 0, stride = 1;
    for (size_t a = r; a-- > 0;) { const int64_t d = t.dims[a]; if (d != 1) off += idx[R - r + a] * stride; stride *= d; }
    return off;
}
bool eval_arith(const ONode& n, const std::vector<const OTensor*>& in, OTensor& y)
{
    static const char* const UN[] = {"Abs", "Neg", "Sin", "Cos", "Exp", "Log", "Sqrt", "Reciprocal", "Floor", "Ceil", "Tanh", "Sigmoid", "Relu", "Not", "Sign"};
    static const char* const BIN[] = {"Add", "Sub", "Mul", "Div", "Pow", "Min", "Max", "Equal", "Less", "Greater", "LessOrEqual", "GreaterOrEqual", "And", "Or"};
    const std::string& op = n.op;
    auto is_in = [&](const char* const* L, size_t cnt) { for (size_t k = 0; k < cnt; ++k) if (op == L[k]) return true; return false; };
    const int64_t LIM = (int64_t)1 << 26;                        // no constant sub-graph of these models comes near 64 M elements
    if (is_in(UN, sizeof(UN) / sizeof(UN[0]))) {
        if (in.size() < 1 || !in[0]) return false;
        const OTensor& x = *in[0];
        const int64_t N = numel(x);
        y.dims = x.dims; y.dtype = x.dtype;
        if (is_floaty(x)) {
            y.f.resize((size_t)N);
            for (int64_t k = 0; k < N; ++k) {
                const float v = x.f[(size_t)k]; float r;
                if (op == "Abs") r = std::fabs(v); else if (op == "Neg") r = -v; else if (op == "Sin") r = std::sin(v); else if (op == "Cos") r = std::cos(v);
                else if (op == "Exp") r = std::exp(v); else if (op == "Log") r = std::log(v); else if (op == "Sqrt") r = std::sqrt(v); else if (op == "Reciprocal") r = 1.0f / v;
                else if (op == "Floor") r = std::floor(v); else if (op == "Ceil") r = std::ceil(v); else if (op == "Tanh") r = std::tanh(v);
                else if (op == "Sigmoid") r = 1.0f / (1.0f + std::exp(-v)); else if (op == "Relu") r = v > 0.0f ? v : 0.0f; else if (op == "Sign") r = (float)((v > 0) - (v < 0));
                else return false;
                y.f[(size_t)k] = r;
            }
        } else {
            y.i64.resize((size_t)N);
            for (int64_t k = 0; k < N; ++k) {
                const int64_t v = x.i64[(size_t)k];
                if (op == "Abs") y.i64[(size_t)k] = v < 0 ? -v : v; else if (op == "Neg") y.i64[(size_t)k] = -v; else if (op == "Not") y.i64[(size_t)k] = !v;
                else if (op == "Sign") y.i64[(size_t)k] = (v > 0) - (v < 0); else if (op == "Relu") y.i64[(size_t)k] = v > 0 ? v : 0; else return false;
            }
        }
        return true;
    }
    if (is_in(BIN, sizeof(BIN) / sizeof(BIN[0])) || op == "Where") {
        const size_t need = op == "Where" ? 3 : 2;
        if (in.size() < need) return false;
        for (auto* t : in) if (!t) return false;
        std::vector<int64_t> od;
        if (!broadcast_dims(in, od)) return false;
        int64_t N = 1;
        for (auto d : od) { if (d < 0 || (d && N > LIM / d)) return false; N *= d; }
        const bool cmp = op == "Equal" || op == "Less" || op == "Greater" || op == "LessOrEqual" || op == "GreaterOrEqual" || op == "And" || op == "Or";
        const size_t v0 = op == "Where" ? 1 : 0;
        bool fl = false;
        for (size_t k = v0; k < in.size(); ++k) fl |= is_floaty(*in[k]);
        y.dims = od; y.dtype = cmp ? 9 : (fl ? 1 : 7);
        if (cmp || !fl) y.i64.resize((size_t)N); else y.f.resize((size_t)N);
        std::vector<int64_t> idx(od.size(), 0);
        for (int64_t k = 0; k < N; ++k) {
            double r;
            if (op == "Where") {
                const bool cnd = getv(*in[0], bc_offset(*in[0], od, idx)) != 0.0;
                r = cnd ? getv(*in[1], bc_offset(*in[1], od, idx)) : getv(*in[2], bc_offset(*in[2], od, idx));
                if (fl) r = (double)(float)r;
            } else {
                // float32 arithmetic exactly as a runtime does it: operands and result in float
                const double a0 = getv(*in[0], bc_offset(*in[0], od, idx));
                r = a0;
                for (size_t q = 1; q < in.size(); ++q) {
                    const double b = getv(*in[q], bc_offset(*in[q], od, idx));
                    if (fl && !cmp) {
                        const float fa = (float)r, fb = (float)b; float fr;
                        if (op == "Add") fr = fa + fb; else if (op == "Sub") fr = fa - fb; else if (op == "Mul") fr = fa * fb; else if (op == "Div") fr = fa / fb;
                        else if (op == "Pow") fr = std::pow(fa, fb); else if (op == "Min") fr = fb < fa ? fb : fa; else if (op == "Max") fr = fb > fa ? fb : fa; else return false;
                        r = fr;
                    } else if (!cmp) {
                        const int64_t ia = (int64_t)r, ib = (int64_t)b; int64_t ir;
                        if (op == "Add") ir = ia + ib; else if (op == "Sub") ir = ia - ib; else if (op == "Mul") ir = ia * ib;
                        else if (op == "Div") { if (ib == 0) return false; ir = ia / ib; }
                        else if (op == "Pow") ir = (int64_t)std::llround(std::pow((double)ia, (double)ib)); else if (op == "Min") ir = ib < ia ? ib : ia; else if (op == "Max") ir = ib > ia ? ib : ia; else return false;
                        r = (double)ir;
                    } else {
                        if (op == "Equal") r = r == b; else if (op == "Less") r = r < b; else if (op == "Greater") r = r > b; else if (op == "LessOrEqual") r = r <= b;
                        else if (op == "GreaterOrEqual") r = r >= b; else if (op == "And") r = (r != 0.0) && (b != 0.0); else r = (r != 0.0) || (b != 0.0);
                        if (q + 1 < in.size()) return false;              // comparisons are binary
                    }
                    if (op != "Min" && op != "Max" && op != "Add" && op != "Mul" && q + 1 < in.size()) return false;      // only these are variadic
                }
            }
            if (cmp || !fl) y.i64[(size_t)k] = (int64_t)r; else y.f[(size_t)k] = (float)r;
            for (size_t a = od.size(); a-- > 0;) { if (++idx[a] < od[a]) break; idx[a] = 0; }
        }
        return true;
    }
    if (op == "Clip") {
        if (in.empty() || !in[0] || !is_floaty(*in[0])) return false;
        float lo = -INFINITY, hi = INFINITY;
        if (const OAttr* a = n.attr("min")) lo = a->f;
        if (const OAttr* a = n.attr("max")) hi = a->f;
        if (in.size() > 1 && in[1]) { if (numel(*in[1]) != 1) return false; lo = (float)getv(*in[1], 0); }
        if (in.size() > 2 && in[2]) { if (numel(*in[2]) != 1) return false; hi = (float)getv(*in[2], 0); }
        y = *in[0];
        for (auto& v : y.f) v = v < lo ? lo : (v > hi ? hi : v);
        return true;
    }
    if (op == "Cast") {
        if (in.empty() || !in[0]) return false;
        const OAttr* a = n.attr("to");
        if (!a) return false;
        const OTensor& x = *in[0];
        const int64_t N = numel(x);
        y.dims = x.dims; y.dtype = (int)a->i;
        if (a->i == 1 || a->i == 11) { y.dtype = 1; y.f.resize((size_t)N); for (int64_t k = 0; k < N; ++k) y.f[(size_t)k] = (float)getv(x, k); }
        else if (a->i == 7 || a->i == 6 || a->i == 9) { y.i64.resize((size_t)N); for (int64_t k = 0; k < N; ++k) { const double v = getv(x, k); y.i64[(size_t)k] = a->i == 9 ? (v != 0.0) : (int64_t)v; } }
        else return false;
        return true;
    }
    if (op == "MatMul" || op == "Gemm") {
        if (in.size() < 2 || !in[0] || !in[1] || !is_floaty(*in[0]) || !is_floaty(*in[1])) return false;
        const OTensor &A = *in[0], &B = *in[1];
        if (A.dims.size() > 2 || B.dims.size() > 2 || A.dims.empty() || B.dims.empty()) return false;
        bool tA = false, tB = false; float alpha = 1.0f, beta = 1.0f;
        if (op == "Gemm") {
            if (const OAttr* a = n.attr("transA")) tA = a->i != 0;
            if (const OAttr* a = n.attr("transB")) tB = a->i != 0;
            if (const OAttr* a = n.attr("alpha")) alpha = a->f;
            if (const OAttr* a = n.attr("beta")) beta = a->f;
        }
        const int64_t ar = A.dims.size() == 2 ? A.dims[0] : 1, ac = A.dims.size() == 2 ? A.dims[1] : A.dims[0];
        const int64_t br = B.dims.size() == 2 ? B.dims[0] : B.dims[0], bc = B.dims.size() == 2 ? B.dims[1] : 1;
        const int64_t M = tA ? ac : ar, K = tA ? ar : ac, K2 = tB ? bc : br, Nn = tB ? br : bc;
        if (K != K2 || M <= 0 || Nn <= 0 || K <= 0 || M > LIM / Nn || M * Nn > LIM / K) return false;      // bounds the work (M N K), not only the output
        if (numel(A) != ar * ac || numel(B) != br * bc) return false;
        y.dtype = 1; y.f.assign((size_t)(M * Nn), 0.0f);
        if (A.dims.size() == 2 && B.dims.size() == 2) y.dims = {M, Nn}; else if (A.dims.size() == 1 && B.dims.size() == 2) y.dims = {Nn}; else if (A.dims.size() == 2) y.dims = {M}; else y.dims = {};
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j < Nn; ++j) {
                float acc = 0.0f;
                for (int64_t k = 0; k < K; ++k) acc += A.f[(size_t)(tA ? k * ac + i : i * ac + k)] * B.f[(size_t)(tB ? j * bc + k : k * bc + j)];
                y.f[(size_t)(i * Nn + j)] = alpha * acc;
            }
        if (op == "Gemm" && in.size() > 2 && in[2]) {
            // C is unidirectionally broadcast to [M, N]: rank <= 2, every dim 1 or the target's, and it must hold what its dims say
            OTensor tgt; tgt.dims = {M, Nn};
            std::vector<const OTensor*> two = {in[2], &tgt};
            std::vector<int64_t> od, idx(2, 0);
            if (in[2]->dims.size() > 2 || numel(*in[2]) <= 0 || !broadcast_dims(two, od) || od.size() != 2 || od[0] != M || od[1] != Nn) return false;
            for (int64_t k = 0; k < M * Nn; ++k) { y.f[(size_t)k] += beta * (float)getv(*in[2], bc_offset(*in[2], od, idx)); if (++idx[1] == Nn) { idx[1] = 0; ++idx[0]; } }
        }
        return true;
    }
    if (op == "Shape") {
        if (in.empty() || !in[0]) return false;
        y.dtype = 7; y.i64 = in[0]->dims; y.dims = {(int64_t)in[0]->dims.size()};
        return true;
    }
    if (op == "Size") { if (in.empty() || !in[0]) return false; y.dtype = 7; y.i64 = {numel(*in[0])}; y.dims = {}; return true; }
    if (op == "ConstantOfShape") {
        if (in.empty() || !in[0] || is_floaty(*in[0])) return false;
        int64_t N = 1;
        for (auto d : in[0]->i64) { if (d < 0 || (d && N > LIM / d)) return false; N *= d; }
        y.dims = in[0]->i64;
        const OAttr* a = n.attr("value");
        if (a && a->has_t && !is_floaty(a->t)) { y.dtype = a->t.dtype ? a->t.dtype : 7; y.i64.assign((size_t)N, a->t.i64.empty() ? 0 : a->t.i64[0]); }
        else { y.dtype = 1; y.f.assign((size_t)N, (a && a->has_t && !a->t.f.empty()) ? a->t.f[0] : 0.0f); }
        return true;
    }
    if (op == "Range") {
        if (in.size() < 3 || !in[0] || !in[1] || !in[2]) return false;
        for (int q = 0; q < 3; ++q) if (numel(*in[q]) != 1) return false;      // scalars (a dims = {0} tensor is sane but empty)
        const double st = getv(*in[0], 0), lim = getv(*in[1], 0), dl = getv(*in[2], 0);
        if (dl == 0.0) return false;
        const double cnt = std::ceil((lim - st) / dl);
        if (!(cnt >= 0) || cnt > (double)LIM) return false;
        const int64_t N = (int64_t)cnt;
        y.dims = {N};
        if (is_floaty(*in[0])) { y.dtype = 1; y.f.resize((size_t)N); for (int64_t k = 0; k < N; ++k) y.f[(size_t)k] = (float)st + (float)k * (float)dl; }
        else { y.dtype = 7; y.i64.resize((size_t)N); for (int64_t k = 0; k < N; ++k) y.i64[(size_t)k] = (int64_t)st + k * (int64_t)dl; }
        return true;
    }
    if (op == "Gather") {
        if (in.size() < 2 || !in[0] || !in[1] || is_floaty(*in[1])) return false;
        const OTensor &x = *in[0], &ix = *in[1];
        const int R = (int)x.dims.size();
        const OAttr* a = n.attr("axis");
        int64_t axis = a ? a->i : 0; if (axis < 0) axis += R;
        if (R == 0 || axis < 0 || axis >= R) return false;
        int64_t outer = 1, inner = 1;
        for (int q = 0; q < axis; ++q) outer *= x.dims[(size_t)q];
        for (int q = (int)axis + 1; q < R; ++q) inner *= x.dims[(size_t)q];
        const int64_t D = x.dims[(size_t)axis], NI = numel(ix);
        if (outer * NI > LIM / std::max<int64_t>(inner, 1)) return false;
        y.dtype = x.dtype; y.dims.clear();
        for (int q = 0; q < axis; ++q) y.dims.push_back(x.dims[(size_t)q]);
        for (auto d : ix.dims) y.dims.push_back(d);
        for (int q = (int)axis + 1; q < R; ++q) y.dims.push_back(x.dims[(size_t)q]);
        const bool fl = is_floaty(x);
        if (fl) y.f.resize((size_t)(outer * NI * inner)); else y.i64.resize((size_t)(outer * NI * inner));
        for (int64_t o = 0; o < outer; ++o)
            for (int64_t k = 0; k < NI; ++k) {
                int64_t g = ix.i64[(size_t)k]; if (g < 0) g += D;
                if (g < 0 || g >= D) return false;
                for (int64_t q = 0; q < inner; ++q) {
                    const size_t src = (size_t)((o * D + g) * inner + q), dst = (size_t)((o * NI + k) * inner + q);
                    if (fl) y.f[dst] = x.f[src]; else y.i64[dst] = x.i64[src];
                }
            }
        return true;
    }
    if (op == "Tile") {
        if (in.size() < 2 || !in[0] || !in[1] || is_floaty(*in[1])) return false;
        const OTensor& x = *in[0];
        const size_t R = x.dims.size();
        if (in[1]->i64.size() != R) return false;
        std::vector<int64_t> od(R);
        int64_t N = 1;
        for (size_t a = 0; a < R; ++a) { const int64_t rp = in[1]->i64[a]; if (rp < 0 || (x.dims[a] && rp > LIM / std::max<int64_t>(x.dims[a], 1))) return false; od[a] = x.dims[a] * rp; if (od[a] && N > LIM / od[a]) return false; N *= od[a]; }
        y.dims = od; y.dtype = x.dtype;
        const bool fl = is_floaty(x);
        if (fl) y.f.resize((size_t)N); else y.i64.resize((size_t)N);
        std::vector<int64_t> idx(R, 0);
        for (int64_t k = 0; k < N; ++k) {
            int64_t off = 0, stride = 1;
            for (size_t a = R; a-- > 0;) { off += (idx[a] % x.dims[a]) * stride; stride *= x.dims[a]; }
            if (fl) y.f[(size_t)k] = x.f[(size_t)off]; else y.i64[(size_t)k] = x.i64[(size_t)off];
            for (size_t a = R; a-- > 0;) { if (++idx[a] < od[a]) break; idx[a] = 0; }
        }
        return true;
    }
    if (op == "Expand") {
        if (in.size() < 2 || !in[0] || !in[1] || is_floaty(*in[1])) return false;
        OTensor shp; shp.dims = in[1]->i64;                           // a tensor that only carries the target dims
        std::vector<const OTensor*> two = {in[0], &shp};
        std::vector<int64_t> od;
        if (!broadcast_dims(two, od)) return false;
        int64_t N = 1;
        for (auto d : od) { if (d < 0 || (d && N > LIM / d)) return false; N *= d; }
        y.dims = od; y.dtype = in[0]->dtype;
        const bool fl = is_floaty(*in[0]);
        if (fl) y.f.resize((size_t)N); else y.i64.resize((size_t)N);
        std::vector<int64_t> idx(od.size(), 0);
        for (int64_t k = 0; k < N; ++k) {
            const int64_t o = bc_offset(*in[0], od, idx);
            if (fl) y.f[(size_t)k] = in[0]->f[(size_t)o]; else y.i64[(size_t)k] = in[0]->i64[(size_t)o];
            for (size_t a = od.size(); a-- > 0;) { if (++idx[a] < od[a]) break; idx[a] = 0; }
        }
        return true;
    }
    return false;
}

void fold_constants(OGraph& g)
{
    for (const ONode& n : g.nodes) {
        if (n.out.empty() || g.init.count(n.out[0])) continue;
        const bool known = n.op == "Slice" || n.op == "Concat" || n.op == "Unsqueeze" || n.op == "Squeeze" || n.op == "Transpose" || n.op == "Identity" ||
                           n.op == "Reshape" || n.op == "Flatten";
        static const char* const ARITH[] = {"Abs", "Neg", "Sin", "Cos", "Exp", "Log", "Sqrt", "Reciprocal", "Floor", "Ceil", "Tanh", "Sigmoid", "Relu", "Not", "Sign",
                                            "Add", "Sub", "Mul", "Div", "Pow", "Min", "Max", "Equal", "Less", "Greater", "LessOrEqual", "GreaterOrEqual", "And", "Or", "Where",
                                            "Clip", "Cast", "MatMul", "Gemm", "Shape", "Size", "ConstantOfShape", "Range", "Gather", "Expand", "Tile"};
        bool arith = false;
        for (auto* a : ARITH) if (n.op == a) arith = true;
        if (!known && !arith) continue;
        std::vector<const OTensor*> in;
        bool all = !n.in.empty();
        for (auto& nm : n.in) {
            if (nm.empty()) { in.push_back(nullptr); continue; }
            auto it = g.init.find(nm);
            if (it == g.init.end() || it->second.external) { all = false; break; }
            in.push_back(&it->second);
        }
        if (!all || !in[0]) continue;
        const OTensor& x = *in[0];
        bool in_ok = true;
        for (auto* t : in) if (t && !sane(*t)) in_ok = false;          // every input, not only the first: Concat copies from all of them
        if (!in_ok) continue;
        if (arith) {
            OTensor ya;
            if (!eval_arith(n, in, ya) || !sane(ya)) continue;
            ya.name = n.out[0];
            g.init[ya.name] = std::move(ya);
            continue;
        }
        OTensor y; y.name = n.out[0]; y.dtype = x.dtype;
        const int R = (int)x.dims.size();
        auto ints_of = [&](size_t k, const char* attr) -> std::vector<int64_t> {
            if (k < in.size() && in[k]) return in[k]->i64;
            const OAttr* a = n.attr(attr);
            return a ? a->ints : std::vector<int64_t>();
        };
        if (n.op == "Identity") { y = x; y.name = n.out[0]; }
        else if (n.op == "Unsqueeze" || n.op == "Squeeze") {
            std::vector<int64_t> axes = ints_of(1, "axes");
            y = x; y.name = n.out[0];
            if (n.op == "Unsqueeze") {
                const int Ro = R + (int)axes.size();
                std::vector<int64_t> od((size_t)Ro, -1);
                for (auto a : axes) { if (a < 0) a += Ro; if (a < 0 || a >= Ro || od[(size_t)a] == 1) { od.clear(); break; } od[(size_t)a] = 1; }   // out of range or duplicated axis
                if (od.empty()) continue;
                size_t q = 0;
                for (auto& d : od) if (d == -1) d = x.dims[q++];
                y.dims = od;
            } else {
                std::vector<int64_t> od;
                bool sq_ok = true;
                for (auto ax : axes) { const int64_t a2 = ax < 0 ? ax + R : ax; if (a2 < 0 || a2 >= R || x.dims[(size_t)a2] != 1) sq_ok = false; }
                if (!sq_ok) continue;
                for (int a = 0; a < R; ++a) {
                    bool drop = axes.empty() ? x.dims[(size_t)a] == 1 : false;
                    for (auto ax : axes) if ((ax < 0 ? ax + R : ax) == a) drop = true;
                    if (!drop) od.push_back(x.dims[(size_t)a]);
                }
                y.dims = od;
            }
        } else if (n.op == "Flatten") {
            const OAttr* aa = n.attr("axis");
            int64_t axis = aa ? aa->i : 1; if (axis < 0) axis += R;
            if (axis < 0 || axis > R) continue;
            int64_t o = 1, i2 = 1;
            for (int q = 0; q < R; ++q) (q < axis ? o : i2) *= x.dims[(size_t)q];
            y = x; y.name = n.out[0]; y.dims = {o, i2};
        } else if (n.op == "Reshape") {
            if (in.size() < 2 || !in[1]) continue;
            std::vector<int64_t> od = in[1]->i64;
            int64_t known_n = 1; int neg = -1;
            bool rs_ok = od.size() <= 8;
            for (size_t a = 0; a < od.size() && rs_ok; ++a) {
                if (od[a] == 0 && a < x.dims.size()) od[a] = x.dims[a];
                if (od[a] == -1) { if (neg >= 0) rs_ok = false; neg = (int)a; }
                else if (od[a] < 0 || od[a] > numel(x) || (od[a] > 0 && known_n > ((int64_t)1 << 40) / od[a])) rs_ok = false;
                else known_n *= od[a];
            }
            if (!rs_ok) continue;
            if (neg >= 0) { if (known_n == 0 || numel(x) % known_n) continue; od[(size_t)neg] = numel(x) / known_n; }
            y = x; y.name = n.out[0]; y.dims = od;
            if (!sane(y)) continue;
        } else if (n.op == "Transpose") {
            const OAttr* pa = n.attr("perm");
            std::vector<int64_t> perm = pa ? pa->ints : std::vector<int64_t>();
            if (perm.empty()) for (int a = R - 1; a >= 0; --a) perm.push_back(a);
            if ((int)perm.size() != R) continue;
            std::vector<char> seen((size_t)R, 0);
            bool pm_ok = true;
            for (auto q : perm) { if (q < 0 || q >= R || seen[(size_t)q]) { pm_ok = false; break; } seen[(size_t)q] = 1; }      // a permutation of 0..R-1
            if (!pm_ok) continue;
            const std::vector<int64_t> xs = strides_of(x.dims);
            std::vector<int64_t> od((size_t)R), st((size_t)R, 0), sp((size_t)R, 1), is((size_t)R);
            for (int a = 0; a < R; ++a) { od[(size_t)a] = x.dims[(size_t)perm[(size_t)a]]; is[(size_t)a] = xs[(size_t)perm[(size_t)a]]; }
            gather_nd(x, od, st, sp, is, y);
        } else if (n.op == "Slice") {
            std::vector<int64_t> starts = ints_of(1, "starts"), ends = ints_of(2, "ends"), axes = ints_of(3, "axes"), steps = ints_of(4, "steps");
            if (starts.size() != ends.size() || starts.empty()) continue;
            if (axes.empty()) for (size_t a = 0; a < starts.size(); ++a) axes.push_back((int64_t)a);
            if (steps.empty()) steps.assign(starts.size(), 1);
            if (axes.size() != starts.size() || steps.size() != starts.size()) continue;
            std::vector<int64_t> od = x.dims, st((size_t)R, 0), sp((size_t)R, 1);
            bool ok = true;
            for (size_t k = 0; k < axes.size(); ++k) {
                int64_t a = axes[k]; if (a < 0) a += R;
                if (a < 0 || a >= R || steps[k] == 0) { ok = false; break; }
                const int64_t D = x.dims[(size_t)a];
                int64_t s0 = starts[k], e0 = ends[k];
                if (steps[k] > 0) {
                    if (s0 < 0) s0 += D; if (e0 < 0) e0 += D;
                    s0 = std::min(std::max<int64_t>(s0, 0), D); e0 = std::min(std::max<int64_t>(e0, 0), D);
                    od[(size_t)a] = e0 > s0 ? (e0 - s0 + steps[k] - 1) / steps[k] : 0;
                } else {
                    // negative step (torch.flip exports as Slice(start = -1, end = INT64_MIN, step = -1)): ONNX clamps start to [0, D - 1], end to [-1, D - 1]
                    if (s0 < 0) s0 += D; if (e0 < 0 && e0 > -((int64_t)1 << 40)) e0 += D;
                    s0 = std::min(std::max<int64_t>(s0, 0), D - 1); e0 = std::min(std::max<int64_t>(e0, -1), D - 1);
                    const int64_t stp = -steps[k];
                    od[(size_t)a] = s0 > e0 ? (s0 - e0 + stp - 1) / stp : 0;
                }
                st[(size_t)a] = s0; sp[(size_t)a] = steps[k];
            }
            if (!ok) continue;
            gather_nd(x, od, st, sp, strides_of(x.dims), y);
        } else if (n.op == "Concat") {
            const OAttr* aa = n.attr("axis");
            int64_t axis = aa ? aa->i : 0; if (axis < 0) axis += R;
            if (axis < 0 || axis >= R) continue;
            bool ok = true; int64_t total = 0;
            for (auto* t : in) {
                if (!t || (int)t->dims.size() != R || is_int(*t) != is_int(x)) { ok = false; break; }
                for (int a = 0; a < R; ++a) if (a != axis && t->dims[(size_t)a] != x.dims[(size_t)a]) ok = false;
                total += t->dims[(size_t)axis];
            }
            if (!ok) continue;
            y.dims = x.dims; y.dims[(size_t)axis] = total;
            int64_t outer = 1, inner = 1;
            for (int a = 0; a < axis; ++a) outer *= x.dims[(size_t)a];
            for (int a = (int)axis + 1; a < R; ++a) inner *= x.dims[(size_t)a];
            const bool ii = is_int(x);
            if (ii) y.i64.resize((size_t)numel(y)); else y.f.resize((size_t)numel(y));
            int64_t off = 0;
            for (auto* t : in) {
                const int64_t w = t->dims[(size_t)axis] * inner;
                if (w == 0) continue;
                for (int64_t o = 0; o < outer; ++o) {
                    if (ii) memcpy(&y.i64[(size_t)(o * total * inner + off)], &t->i64[(size_t)(o * w)], (size_t)w * 8);
                    else memcpy(&y.f[(size_t)(o * total * inner + off)], &t->f[(size_t)(o * w)], (size_t)w * 4);
                }
                off += w;
            }
        }
        g.init[y.name] = std::move(y);
    }
}

int load_onnx(const char* path, OGraph& g, std::string& err)
{
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open model file: ") + path; return SD_ERR_MODEL; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(sz > 0 ? sz : 0));
    if (sz <= 0 || fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); err = std::string("cannot read model file: ") + path; return SD_ERR_MODEL; }
    fclose(f);
    PB pb(buf.data(), buf.size());
    int fld, wt; bool got = false;
    while (pb.tag(fld, wt)) {
        if (fld == 7 && wt == 2) { if (!parse_graph(pb.sub(), g)) { err = std::string("malformed ONNX graph in ") + path; return SD_ERR_MODEL; } got = true; }
        else pb.skip(wt);
    }
    if (!pb.ok || !got) { err = std::string("not an ONNX ModelProto: ") + path; return SD_ERR_MODEL; }
    fold_constants(g);
    return SD_OK;
}

const OTensor* weight_of(const OGraph& g, const std::string& name, std::string& err, const char* what)
{
    auto it = g.init.find(name);
    if (it == g.init.end()) { err = std::string(what) + ": input '" + name + "' is not a constant (weights must be initializers / folded constants)"; return nullptr; }
    if (it->second.external) { err = std::string(what) + ": external tensor data is not supported"; return nullptr; }
    if (it->second.f.empty()) { err = std::string(what) + ": tensor '" + name + "' holds no float data"; return nullptr; }
    return &it->second;
}

void put(Pack& p, const std::string& name, const std::vector<int64_t>& dims, const std::vector<float>& data)
{
    PackTensor t; t.dims = dims; t.data = data; p[name] = std::move(t);
}

// ---- PyanNet (segment2.onnx)
int seg_from_onnx(const OGraph& g, Pack& p, std::string& err)
{
    std::vector<const ONode*> inorm, conv, lstm, mm;
    for (auto& n : g.nodes) {
        if (!n.out.empty() && g.init.count(n.out[0])) continue;          // part of a constant sub-graph (e.g. the filter bank's own MatMuls): folded, not a layer
        if (n.op == "InstanceNormalization") inorm.push_back(&n);
        else if (n.op == "Conv") conv.push_back(&n);
        else if (n.op == "LSTM") lstm.push_back(&n);
        else if (n.op == "MatMul" || n.op == "Gemm") mm.push_back(&n);
    }
    if (inorm.size() != 4 || conv.size() != 3 || lstm.size() != 4 || mm.size() != 3) {
        char b[256]; snprintf(b, sizeof(b), "segmentation graph: expected 4 InstanceNormalization / 3 Conv / 4 LSTM / 3 MatMul, found %zu / %zu / %zu / %zu",
                              inorm.size(), conv.size(), lstm.size(), mm.size());
        err = b; return SD_ERR_MODEL;
    }
    const char* inn[4] = {"sincnet.wav_norm", "sincnet.norm0", "sincnet.norm1", "sincnet.norm2"};
    for (int i = 0; i < 4; ++i) {
        const OTensor *s = weight_of(g, inorm[i]->in[1], err, "InstanceNormalization scale"), *b = weight_of(g, inorm[i]->in[2], err, "InstanceNormalization bias");
        if (!s || !b) return SD_ERR_MODEL;
        put(p, std::string(inn[i]) + ".weight", {(int64_t)s->f.size()}, s->f);
        put(p, std::string(inn[i]) + ".bias", {(int64_t)b->f.size()}, b->f);
    }
    for (int i = 0; i < 3; ++i) {
        const OTensor* w = weight_of(g, conv[i]->in[1], err, "Conv weight");
        if (!w || w->dims.size() != 3) { if (w) err = "Conv weight is not 3-D"; return SD_ERR_MODEL; }
        put(p, "sincnet.conv" + std::to_string(i) + ".weight", w->dims, w->f);
        if (conv[i]->in.size() > 2 && !conv[i]->in[2].empty()) {
            const OTensor* b = weight_of(g, conv[i]->in[2], err, "Conv bias");
            if (!b) return SD_ERR_MODEL;
            put(p, "sincnet.conv" + std::to_string(i) + ".bias", {(int64_t)b->f.size()}, b->f);
        } else if (i > 0) put(p, "sincnet.conv" + std::to_string(i) + ".bias", {w->dims[0]}, std::vector<float>((size_t)w->dims[0], 0.0f));
    }
    // LSTM: W [2][4H][in], R [2][4H][H], B [2][8H]; ONNX gate order i,o,f,c -> PyTorch i,f,g,o
    for (int l = 0; l < 4; ++l) {
        const ONode& n = *lstm[l];
        if (n.in.size() < 4) { err = "LSTM node without W/R/B"; return SD_ERR_MODEL; }
        const OTensor *W = weight_of(g, n.in[1], err, "LSTM W"), *R = weight_of(g, n.in[2], err, "LSTM R"), *B = weight_of(g, n.in[3], err, "LSTM B");
        if (!W || !R || !B) return SD_ERR_MODEL;
        if (W->dims.size() != 3 || W->dims[0] != 2 || R->dims.size() != 3 || R->dims[0] != 2) { err = "LSTM is not bidirectional [2,4H,*]"; return SD_ERR_MODEL; }
        const int64_t H4 = W->dims[1], H = H4 / 4, nin = W->dims[2];
        if (R->dims[1] != H4 || R->dims[2] != H || (int64_t)B->f.size() != 2 * 2 * H4) { err = "LSTM W/R/B shapes disagree"; return SD_ERR_MODEL; }
        const int src_of_dst[4] = {0, 2, 3, 1};        // pytorch gate k (i,f,g,o) <- onnx gate (i,o,f,c) index
        for (int d = 0; d < 2; ++d) {
            std::vector<float> wih((size_t)H4 * nin), whh((size_t)H4 * H), bih((size_t)H4), bhh((size_t)H4);
            for (int k = 0; k < 4; ++k)
                for (int64_t r = 0; r < H; ++r) {
                    const int64_t dst = k * H + r, src = src_of_dst[k] * H + r;
                    memcpy(&wih[(size_t)dst * nin], &W->f[((size_t)d * H4 + src) * nin], (size_t)nin * 4);
                    memcpy(&whh[(size_t)dst * H], &R->f[((size_t)d * H4 + src) * H], (size_t)H * 4);
                    bih[(size_t)dst] = B->f[(size_t)d * 2 * H4 + src];
                    bhh[(size_t)dst] = B->f[(size_t)d * 2 * H4 + H4 + src];
                }
            const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
            put(p, "lstm.weight_ih" + sfx, {H4, nin}, wih);
            put(p, "lstm.weight_hh" + sfx, {H4, H}, whh);
            put(p, "lstm.bias_ih" + sfx, {H4}, bih);
            put(p, "lstm.bias_hh" + sfx, {H4}, bhh);
        }
    }
    // Linear layers: MatMul(x, W^T [in,out]) followed by Add(bias); Gemm carries both
    const char* ln[3] = {"linear.0", "linear.1", "classifier"};
    for (int i = 0; i < 3; ++i) {
        const ONode& n = *mm[i];
        const OTensor* w = nullptr; bool transposed = true;        // MatMul constant is [in,out]
        for (size_t k = 0; k < n.in.size() && k < 2; ++k) { auto it = g.init.find(n.in[k]); if (it != g.init.end() && it->second.dims.size() == 2) w = &it->second; }
        if (!w || w->f.empty()) { err = "linear layer weight is not a constant 2-D tensor"; return SD_ERR_MODEL; }
        if (n.op == "Gemm") { const OAttr* tb = n.attr("transB"); transposed = !(tb && tb->i == 1); }
        const int64_t d0 = w->dims[0], d1 = w->dims[1];
        const int64_t out = transposed ? d1 : d0, in = transposed ? d0 : d1;
        std::vector<float> wt((size_t)out * in);
        for (int64_t o = 0; o < out; ++o) for (int64_t q = 0; q < in; ++q) wt[(size_t)o * in + q] = transposed ? w->f[(size_t)q * d1 + o] : w->f[(size_t)o * d1 + q];
        put(p, std::string(ln[i]) + ".weight", {out, in}, wt);
        const OTensor* b = nullptr;
        if (n.op == "Gemm" && n.in.size() > 2) b = weight_of(g, n.in[2], err, "Gemm bias");
        else {
            for (auto& a : g.nodes) {          // the Add consuming this MatMul's output
                if (a.op != "Add" || a.in.size() != 2) continue;
                for (int k = 0; k < 2; ++k) if (a.in[k] == n.out[0]) { auto it = g.init.find(a.in[1 - k]); if (it != g.init.end()) b = &it->second; }
            }
        }
        if (!b || (int64_t)b->f.size() != out) { err = "linear layer bias not found"; return SD_ERR_MODEL; }
        put(p, std::string(ln[i]) + ".bias", {out}, b->f);
    }
    return SD_OK;
}

// ---- ECAPA-TDNN front end + body (emd4.onnx)
int emb_from_onnx(const OGraph& g, Pack& p, std::string& err)
{
    std::vector<const ONode*> conv, bn;
    const OTensor* mel = nullptr;
    for (auto& n : g.nodes) {
        if (!n.out.empty() && g.init.count(n.out[0])) continue;          // folded constant sub-graph
        if (n.op == "Conv") conv.push_back(&n);
        else if (n.op == "BatchNormalization") bn.push_back(&n);
        else if (n.op == "MatMul" && !mel)
            for (auto& nm : n.in) { auto it = g.init.find(nm); if (it != g.init.end() && it->second.dims.size() == 2 && it->second.dims[0] == SD_NBINS && it->second.dims[1] == SD_NMELS) mel = &it->second; }
    }
    if (!mel) { err = "embedding graph: no MatMul with a constant [201,80] mel filterbank"; return SD_ERR_MODEL; }
    if (conv.size() != 38 || bn.size() != 31) {
        char b[200]; snprintf(b, sizeof(b), "embedding graph: expected 38 Conv / 31 BatchNormalization (ECAPA-TDNN C=1024), found %zu / %zu", conv.size(), bn.size());
        err = b; return SD_ERR_MODEL;
    }
    put(p, "fbank.matrix", mel->dims, mel->f);
    std::vector<std::string> cn, bnn;
    cn.push_back("blocks.0.conv"); bnn.push_back("blocks.0.norm");
    for (int b = 1; b <= 3; ++b) {
        const std::string pre = "blocks." + std::to_string(b);
        cn.push_back(pre + ".tdnn1.conv"); bnn.push_back(pre + ".tdnn1.norm");
        for (int i = 0; i < 7; ++i) { cn.push_back(pre + ".res2net." + std::to_string(i) + ".conv"); bnn.push_back(pre + ".res2net." + std::to_string(i) + ".norm"); }
        cn.push_back(pre + ".tdnn2.conv"); bnn.push_back(pre + ".tdnn2.norm");
        cn.push_back(pre + ".se.conv1"); cn.push_back(pre + ".se.conv2");
    }
    cn.push_back("mfa.conv"); bnn.push_back("mfa.norm");
    cn.push_back("asp.tdnn.conv"); bnn.push_back("asp.tdnn.norm");
    cn.push_back("asp.conv"); bnn.push_back("asp_bn");
    cn.push_back("fc");
    for (size_t i = 0; i < conv.size(); ++i) {
        const OTensor* w = weight_of(g, conv[i]->in[1], err, "Conv weight");
        if (!w || w->dims.size() != 3) { if (w) err = "Conv weight is not 3-D"; return SD_ERR_MODEL; }
        put(p, cn[i] + ".weight", w->dims, w->f);
        std::vector<float> bias((size_t)w->dims[0], 0.0f);
        if (conv[i]->in.size() > 2 && !conv[i]->in[2].empty()) { const OTensor* b = weight_of(g, conv[i]->in[2], err, "Conv bias"); if (!b) return SD_ERR_MODEL; bias = b->f; }
        put(p, cn[i] + ".bias", {w->dims[0]}, bias);
    }
    for (size_t i = 0; i < bn.size(); ++i) {
        if (bn[i]->in.size() < 5) { err = "BatchNormalization without 5 inputs"; return SD_ERR_MODEL; }
        const char* part[4] = {".weight", ".bias", ".running_mean", ".running_var"};
        for (int k = 0; k < 4; ++k) {
            const OTensor* t = weight_of(g, bn[i]->in[1 + k], err, "BatchNormalization parameter");
            if (!t) return SD_ERR_MODEL;
            put(p, bnn[i] + part[k], {(int64_t)t->f.size()}, t->f);
        }
        const OAttr* eps = bn[i]->attr("epsilon");
        if (eps && fabsf(eps->f - 1e-5f) > 1e-9f) { err = "BatchNormalization epsilon != 1e-5"; return SD_ERR_MODEL; }
    }
    // shape sanity for the architecture the kernels implement
    if (p["blocks.0.conv.weight"].dims[1] != SD_NMELS || p["fc.weight"].dims[0] != SD_EMB_DIM) { err = "embedding graph: unexpected block0 / fc shapes"; return SD_ERR_MODEL; }
    return SD_OK;
}

int save_pack(const char* path, const Pack& p, std::string& err)
{
    FILE* f = fopen(path, "wb");
    if (!f) { err = std::string("cannot write ") + path; return SD_ERR_ARG; }
    const uint32_t cnt = (uint32_t)p.size();
    fwrite("SDW1", 1, 4, f); fwrite(&cnt, 4, 1, f);
    for (auto& kv : p) {
        const uint16_t nl = (uint16_t)kv.first.size(); const uint8_t nd = (uint8_t)kv.second.dims.size();
        fwrite(&nl, 2, 1, f); fwrite(kv.first.data(), 1, nl, f); fwrite(&nd, 1, 1, f);
        if (nd) fwrite(kv.second.dims.data(), 8, nd, f);
        if (!kv.second.data.empty()) fwrite(kv.second.data.data(), 4, kv.second.data.size(), f);
    }
    fclose(f);
    return SD_OK;
}

}  // namespace

// kind: 0 = segmentation (PyanNet), 1 = embedding (ECAPA-TDNN)
int load_model_any(const char* path, int kind, Pack& out, std::string& err)
{
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open model file: ") + path; return SD_ERR_MODEL; }
    char magic[4] = {0, 0, 0, 0};
    const size_t got = fread(magic, 1, 4, f);
    fclose(f);
    if (got == 4 && memcmp(magic, "SDW1", 4) == 0) return load_pack(path, out, err);
    OGraph g;
    int rc = load_onnx(path, g, err);
    if (rc) return rc;
    rc = kind == 0 ? seg_from_onnx(g, out, err) : emb_from_onnx(g, out, err);
    if (rc) err = std::string(path) + ": " + err;
    return rc;
}

static std::string g_convert_err;
extern "C" const char* sd_convert_error(void) { return g_convert_err.c_str(); }
extern "C" int sd_convert_onnx(const char* onnx_path, int kind, const char* out_sdw_path)
{
    g_convert_err.clear();
    if (!onnx_path || !out_sdw_path || (kind != 0 && kind != 1)) { g_convert_err = "bad argument"; return SD_ERR_ARG; }
    Pack p;
    int rc = load_model_any(onnx_path, kind, p, g_convert_err);
    if (rc) return rc;
    return save_pack(out_sdw_path, p, g_convert_err);
}
