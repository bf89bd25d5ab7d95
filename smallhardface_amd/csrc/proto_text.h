// Minimal protobuf text-format reader (prototxt) and binary wire-format reader
// (.caffemodel) -- there is no protoc / libprotobuf in the image.  Covers what
// Net::Net (caffe/src/caffe/net.cpp:28-43, util/upgrade_proto.cpp:86,966-1000) and
// CopyTrainedLayersFrom (net.cpp:733-768) need for the detector's graphs.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace shf {

struct PMsg;
struct PField {
  std::string name;
  std::string scalar;  // text of a scalar value (unquoted)
  bool quoted = false;
  std::shared_ptr<PMsg> msg;  // non-null for a sub-message
};
struct PMsg {
  std::vector<PField> fields;
  std::vector<const PField*> all(const std::string& n) const {
    std::vector<const PField*> r;
    for (auto& f : fields)
      if (f.name == n) r.push_back(&f);
    return r;
  }
  const PField* get(const std::string& n) const {
    for (auto& f : fields)
      if (f.name == n) return &f;
    return nullptr;
  }
  const PMsg* sub(const std::string& n) const {
    auto f = get(n);
    return (f && f->msg) ? f->msg.get() : nullptr;
  }
  std::string str(const std::string& n, const std::string& d = "") const {
    auto f = get(n);
    return (f && !f->msg) ? f->scalar : d;
  }
  long num(const std::string& n, long d) const {
    auto f = get(n);
    return (f && !f->msg) ? std::strtol(f->scalar.c_str(), nullptr, 10) : d;
  }
  double real(const std::string& n, double d) const {
    auto f = get(n);
    return (f && !f->msg) ? std::strtod(f->scalar.c_str(), nullptr) : d;
  }
};

class TextParser {
 public:
  explicit TextParser(const std::string& s) : s_(s) {}
  std::shared_ptr<PMsg> parse() {
    auto m = message(0);
    return m;
  }
  // protobuf's own text / wire parsers stop at a recursion depth of 100 (io::CodedInputStream default_recursion_limit_);
  // without a limit a file of 10^6 '{' overflows the stack of the process that only wanted to load a model
  static constexpr int kMaxDepth = 100;

 private:
  const std::string& s_;
  size_t i_ = 0;
  int depth_ = 0;
  void skip() {
    for (;;) {
      while (i_ < s_.size() && (isspace((unsigned char)s_[i_]) || s_[i_] == ',' || s_[i_] == ';')) ++i_;
      if (i_ < s_.size() && s_[i_] == '#') {
        while (i_ < s_.size() && s_[i_] != '\n') ++i_;
        continue;
      }
      break;
    }
  }
  static bool atomch(char c) {
    return !(isspace((unsigned char)c) || strchr("{}:[],;<>\"'#", c));
  }
  std::string atom() {
    size_t b = i_;
    while (i_ < s_.size() && atomch(s_[i_])) ++i_;
    if (b == i_) throw std::runtime_error("prototxt: unexpected character near offset " + std::to_string(i_));
    return s_.substr(b, i_ - b);
  }
  std::string quoted() {
    const char q = s_[i_++];
    std::string out;
    while (i_ < s_.size() && s_[i_] != q) {
      if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
        ++i_;
        const char c = s_[i_++];
        out.push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c);
      } else {
        out.push_back(s_[i_++]);
      }
    }
    if (i_ >= s_.size()) throw std::runtime_error("prototxt: unterminated string");
    ++i_;
    return out;
  }
  std::shared_ptr<PMsg> message(char close) {
    auto m = std::make_shared<PMsg>();
    for (;;) {
      skip();
      if (i_ >= s_.size()) {
        if (close) throw std::runtime_error("prototxt: unterminated message");
        return m;
      }
      if (close && s_[i_] == close) {
        ++i_;
        return m;
      }
      PField f;
      f.name = atom();
      skip();
      if (i_ < s_.size() && s_[i_] == ':') {
        ++i_;
        skip();
      }
      if (i_ >= s_.size()) throw std::runtime_error("prototxt: value expected after " + f.name);
      if (s_[i_] == '{' || s_[i_] == '<') {
        const char c = s_[i_] == '{' ? '}' : '>';
        ++i_;
        if (++depth_ > kMaxDepth) throw std::runtime_error("prototxt: messages nested deeper than " + std::to_string(kMaxDepth));
        f.msg = message(c);
        --depth_;
        m->fields.push_back(f);
      } else if (s_[i_] == '[') {
        ++i_;
        for (;;) {
          skip();
          if (i_ >= s_.size()) throw std::runtime_error("prototxt: unterminated list");
          if (s_[i_] == ']') {
            ++i_;
            break;
          }
          PField g;
          g.name = f.name;
          if (s_[i_] == '"' || s_[i_] == '\'') {
            g.scalar = quoted();
            g.quoted = true;
          } else {
            g.scalar = atom();
          }
          m->fields.push_back(g);
        }
      } else if (s_[i_] == '"' || s_[i_] == '\'') {
        f.quoted = true;
        f.scalar = quoted();
        for (;;) {  // adjacent literals concatenate
          skip();
          if (i_ < s_.size() && (s_[i_] == '"' || s_[i_] == '\'')) f.scalar += quoted();
          else break;
        }
        m->fields.push_back(f);
      } else {
        f.scalar = atom();
        m->fields.push_back(f);
      }
    }
  }
};

// ---- binary wire format (only what .caffemodel weight loading needs) -----------
struct WireBlob {
  std::vector<int64_t> shape;
  std::vector<float> data;
};
struct WireLayer {
  std::string name, type;
  std::vector<WireBlob> blobs;
};

class WireReader {
 public:
  WireReader(const uint8_t* p, size_t n) : p_(p), e_(p + n) {}
  bool done() const { return p_ >= e_; }
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    while (p_ < e_) {
      const uint8_t b = *p_++;
      v |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) return v;
      sh += 7;
      if (sh > 63) break;
    }
    throw std::runtime_error("caffemodel: bad varint");
  }
  // returns field number, sets wire type
  uint32_t tag(int& wt) {
    const uint64_t t = varint();
    wt = (int)(t & 7);
    return (uint32_t)(t >> 3);
  }
  WireReader sub() {
    const uint64_t n = varint();
    if ((uint64_t)(e_ - p_) < n) throw std::runtime_error("caffemodel: truncated");
    WireReader r(p_, (size_t)n);
    p_ += n;
    return r;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: adv(8); break;
      case 2: { const uint64_t n = varint(); adv((size_t)n); break; }
      case 5: adv(4); break;
      default: throw std::runtime_error("caffemodel: unsupported wire type");
    }
  }
  float f32() {
    float f;
    if (e_ - p_ < 4) throw std::runtime_error("caffemodel: truncated");
    memcpy(&f, p_, 4);
    p_ += 4;
    return f;
  }
  std::string bytes() {
    const uint64_t n = varint();
    if ((uint64_t)(e_ - p_) < n) throw std::runtime_error("caffemodel: truncated");
    std::string s((const char*)p_, (size_t)n);
    p_ += n;
    return s;
  }
  size_t left() const { return (size_t)(e_ - p_); }

 private:
  void adv(size_t n) {
    if ((size_t)(e_ - p_) < n) throw std::runtime_error("caffemodel: truncated");
    p_ += n;
  }
  const uint8_t* p_;
  const uint8_t* e_;
};

// BlobProto (caffe.proto:10-22): shape=7{dim=1}, data=5 (packed float), legacy num/channels/height/width=1..4
inline WireBlob read_blob(WireReader r) {
  WireBlob b;
  int64_t legacy[4] = {0, 0, 0, 0};
  bool has_legacy = false;
  while (!r.done()) {
    int wt;
    const uint32_t f = r.tag(wt);
    if (f == 7 && wt == 2) {
      WireReader s = r.sub();
      while (!s.done()) {
        int w2;
        const uint32_t g = s.tag(w2);
        if (g == 1 && w2 == 2) {
          WireReader d = s.sub();
          while (!d.done()) b.shape.push_back((int64_t)d.varint());
        } else if (g == 1 && w2 == 0) {
          b.shape.push_back((int64_t)s.varint());
        } else {
          s.skip(w2);
        }
      }
    } else if (f == 5 && wt == 2) {
      WireReader d = r.sub();
      b.data.reserve(d.left() / 4);
      while (!d.done()) b.data.push_back(d.f32());
    } else if (f == 5 && wt == 5) {
      b.data.push_back(r.f32());
    } else if (f >= 1 && f <= 4 && wt == 0) {
      legacy[f - 1] = (int64_t)r.varint();
      has_legacy = true;
    } else {
      r.skip(wt);
    }
  }
  if (b.shape.empty() && has_legacy) b.shape.assign(legacy, legacy + 4);
  // Blob::Reshape (blob.cpp:23-51): every dim >= 0 and the running count <= INT_MAX; Blob::FromProto (blob.cpp:430-…):
  // the float data, when present, has exactly count elements
  if (b.shape.size() > 32) throw std::runtime_error("caffemodel: blob with more than 32 axes");
  int64_t count = 1;
  for (int64_t d : b.shape) {
    if (d < 0) throw std::runtime_error("caffemodel: negative blob dimension");
    if (d != 0 && count > (int64_t)INT32_MAX / d) throw std::runtime_error("caffemodel: blob size exceeds INT_MAX");
    count *= d;
  }
  if (!b.shape.empty() && !b.data.empty() && (int64_t)b.data.size() != count)
    throw std::runtime_error("caffemodel: blob data does not match its shape");
  return b;
}

// NetParameter.layer = 100 (LayerParameter: name=1, type=2, blobs=7); caffe.proto:64-96,306-…
inline std::vector<WireLayer> parse_caffemodel(const uint8_t* data, size_t size) {
  WireReader r(data, size);
  std::vector<WireLayer> out;
  while (!r.done()) {
    int wt;
    const uint32_t fno = r.tag(wt);
    if (fno == 100 && wt == 2) {
      WireReader lr = r.sub();
      WireLayer L;
      while (!lr.done()) {
        int w2;
        const uint32_t g = lr.tag(w2);
        if (g == 1 && w2 == 2) L.name = lr.bytes();
        else if (g == 2 && w2 == 2) L.type = lr.bytes();
        else if (g == 7 && w2 == 2) L.blobs.push_back(read_blob(lr.sub()));
        else lr.skip(w2);
      }
      out.push_back(std::move(L));
    } else if (fno == 2 && wt == 2) {
      // legacy V1LayerParameter (NetParameter.layers = 2: name=4, type=5 enum, blobs=6; caffe.proto:1247-1290),
      // what UpgradeV1Net (upgrade_proto.cpp) turns into `layer` before CopyTrainedLayersFrom matches by name
      WireReader lr = r.sub();
      WireLayer L;
      L.type = "V1";
      while (!lr.done()) {
        int w2;
        const uint32_t g = lr.tag(w2);
        if (g == 4 && w2 == 2) L.name = lr.bytes();
        else if (g == 6 && w2 == 2) L.blobs.push_back(read_blob(lr.sub()));
        else lr.skip(w2);
      }
      out.push_back(std::move(L));
    } else {
      r.skip(wt);
    }
  }
  return out;
}

inline std::vector<WireLayer> read_caffemodel(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("could not open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string buf = ss.str();
  return parse_caffemodel((const uint8_t*)buf.data(), buf.size());
}

}  // namespace shf
