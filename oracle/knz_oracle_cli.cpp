// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// Command-line driver over the CPU restatement: knz_oracle -c|-d -i in -o out [-t T] [-e E] [-b SIZE] [-x 32|64] [-j N]
// Flag meaning follows v2/app/Kanzi.go:195-920 for the subset used by BASELINE.json's configs.
#include "stream.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace knzo;

static uint64_t transformId(const std::string& tok) {
    if (tok == "NONE") return T_NONE; if (tok == "BWT") return T_BWT; if (tok == "LZ") return T_LZ;
    if (tok == "LZX") return T_LZX; if (tok == "ZRLT") return T_ZRLT; if (tok == "MTFT") return T_MTFT;
    if (tok == "RANK") return T_RANK; if (tok == "SRT") return T_SRT; if (tok == "LZP") return T_LZP; if (tok == "UTF") return T_UTF; if (tok == "TEXT") return T_TEXT;
    fprintf(stderr, "unknown transform %s\n", tok.c_str()); exit(2);
}
// Factory.go:289-328 GetType
static uint64_t transformType(const std::string& name) {
    uint64_t res = 0; int shift = 42; size_t pos = 0;
    while (pos <= name.size()) {
        size_t e = name.find('+', pos); if (e == std::string::npos) e = name.size();
        uint64_t t = transformId(name.substr(pos, e - pos));
        if (t != T_NONE) { res |= t << shift; shift -= 6; }
        pos = e + 1;
    }
    return res;
}
static uint32_t entropyType(const std::string& n) {
    if (n == "NONE") return E_NONE; if (n == "HUFFMAN") return E_HUFFMAN; if (n == "ANS0") return E_ANS0;
    if (n == "ANS1") return E_ANS1; if (n == "FPAQ") return E_FPAQ;
    fprintf(stderr, "unknown entropy %s\n", n.c_str()); exit(2);
}
static size_t parseSize(std::string s) { // Kanzi.go:708-733
    size_t mul = 1; char c = s.empty() ? 0 : (char)toupper(s.back());
    if (c == 'K') mul = 1024; else if (c == 'M') mul = 1024 * 1024; else if (c == 'G') mul = 1024 * 1024 * 1024;
    if (mul != 1) s.pop_back();
    return (size_t)atoll(s.c_str()) * mul;
}
static std::vector<uint8_t> readFile(const char* p) {
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n); if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(1); }
    fclose(f); return v;
}
int main(int argc, char** argv) {
    bool comp = true; std::string in, out, t = "NONE", e = "NONE"; size_t bs = 4 << 20; int x = 0, j = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto nxt = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return std::string(argv[++i]); };
        if (a == "-c") comp = true; else if (a == "-d") comp = false; else if (a == "-i") in = nxt(); else if (a == "-o") out = nxt();
        else if (a == "-t") t = nxt(); else if (a == "-e") e = nxt(); else if (a == "-b") bs = parseSize(nxt());
        else if (a == "-x") x = atoi(nxt().c_str()); else if (a == "-j") j = atoi(nxt().c_str());
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "usage: knz_oracle -c|-d -i in -o out [-t T] [-e E] [-b size] [-x 32|64] [-j N]\n"); return 2; }
    std::vector<uint8_t> src = readFile(in.c_str()), dst;
    auto t0 = std::chrono::steady_clock::now();
    try {
        if (comp) compressStream(src.data(), src.size(), transformType(t), entropyType(e), bs, x, j, (int64_t)src.size(), dst);
        else decompressStream(src.data(), src.size(), j, dst);
    } catch (const KnzError& err) { fprintf(stderr, "error %d: %s\n", err.code, err.what()); return err.code; }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    FILE* f = fopen(out.c_str(), "wb"); if (!f) { perror(out.c_str()); return 1; }
    if (!dst.empty()) fwrite(dst.data(), 1, dst.size(), f); fclose(f);
    size_t raw = comp ? src.size() : dst.size();
    fprintf(stderr, "%s %zu -> %zu bytes in %.1f ms (%.1f MB/s, %d threads)\n", comp ? "compressed" : "decompressed", src.size(), dst.size(), ms, raw / 1e3 / ms, j);
    return 0;
}
