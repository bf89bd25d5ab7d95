// Host-only harness around csrc/proto_text.h for tests/test_parser_robustness.py: built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined
// and fed hostile prototxt / .caffemodel bytes.  Every input is copied into an EXACT-SIZE heap buffer first, so a read
// one byte past the input is a sanitizer report, not luck.
//   parser_harness model  FILE     parse FILE as a .caffemodel           -> "OK layers=N floats=M" | "ERR <message>"
//   parser_harness text   FILE     parse FILE as a prototxt              -> "OK fields=N"          | "ERR <message>"
//   parser_harness truncs FILE     every prefix of FILE as a .caffemodel -> one line per prefix length
// Exit code 0 in all three cases (a parse error is a RESULT here); anything else is a crash or a sanitizer abort.
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>

#include "proto_text.h"

static std::string slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static size_t count_fields(const shf::PMsg& m) {
  size_t n = m.fields.size();
  for (auto& f : m.fields)
    if (f.msg) n += count_fields(*f.msg);
  return n;
}

static void model(const uint8_t* p, size_t n, const char* prefix) {
  std::unique_ptr<uint8_t[]> exact(new uint8_t[n ? n : 1]);
  if (n) memcpy(exact.get(), p, n);
  try {
    auto layers = shf::parse_caffemodel(exact.get(), n);
    size_t floats = 0;
    for (auto& L : layers)
      for (auto& b : L.blobs) floats += b.data.size();
    printf("%sOK layers=%zu floats=%zu\n", prefix, layers.size(), floats);
  } catch (const std::exception& e) {
    printf("%sERR %s\n", prefix, e.what());
  }
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  const std::string mode = argv[1], buf = slurp(argv[2]);
  if (mode == "model") {
    model((const uint8_t*)buf.data(), buf.size(), "");
  } else if (mode == "truncs") {
    for (size_t n = 0; n <= buf.size(); ++n) {
      char pre[32];
      snprintf(pre, sizeof pre, "%zu ", n);
      model((const uint8_t*)buf.data(), n, pre);
    }
  } else if (mode == "text") {
    const std::string exact(buf.data(), buf.size());
    try {
      shf::TextParser tp(exact);
      auto m = tp.parse();
      printf("OK fields=%zu\n", count_fields(*m));
    } catch (const std::exception& e) {
      printf("ERR %s\n", e.what());
    }
  } else {
    return 2;
  }
  return 0;
}
