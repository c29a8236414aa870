// parquet_emu_main.cpp -- stand-alone driver of the CPU harness, built with -fsanitize=address,undefined by the tests: every column
// of every file given (well-formed and deliberately corrupted ones) goes through the product's reader; *.snappy files hold
// [u32 uncompressed length][raw Snappy stream] and go through the wavefront rounds, *.zst files [u32 uncompressed length][zstd frames] and go through
// the index pass + the entropy / execute bodies.  Any out-of-bounds access aborts the process.
#include "parquet_emu.cpp"

int main(int argc, char** argv) {
  int ok = 0, unsupported = 0, invalid = 0;
  for (int a = 1; a < argc; a++) {
    std::string path = argv[a];
    if (path.size() > 7 && path.substr(path.size() - 7) == ".snappy") {
      FILE* f = fopen(path.c_str(), "rb");
      if (!f) return 2;
      std::vector<uint8_t> buf;
      uint8_t tmp[4096];
      size_t n;
      while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
      fclose(f);
      if (buf.size() < 4) return 2;
      uint32_t n_out;
      memcpy(&n_out, buf.data(), 4);
      // exact-size heap blocks: the sanitizer sees any access past the stream or past the output
      std::vector<uint8_t> out(n_out);
      for (int order = 0; order < 2; order++) {
        uint32_t rounds = 0;
        int e = pqemu_snappy(buf.data() + 4, (uint32_t)buf.size() - 4, out.data(), n_out, order, &rounds);
        (e ? invalid : ok)++;
      }
      continue;
    }
    if (path.size() > 4 && path.substr(path.size() - 4) == ".zst") {
      FILE* f = fopen(path.c_str(), "rb");
      if (!f) return 2;
      std::vector<uint8_t> buf;
      uint8_t tmp[4096];
      size_t n;
      while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
      fclose(f);
      if (buf.size() < 4) return 2;
      uint32_t n_out;
      memcpy(&n_out, buf.data(), 4);
      std::vector<uint8_t> out(n_out);
      for (int order = 0; order < 2; order++) {
        int e = pqemu_zstd(buf.data() + 4, (uint32_t)buf.size() - 4, out.data(), n_out, order, nullptr);
        (e ? invalid : ok)++;
      }
      continue;
    }
    int n_cols = 0, n_rg = 0;
    try {
      std::unique_ptr<File> f = open_file(path);
      n_cols = (int)f->md.leaves.size();
      n_rg = (int)f->md.row_groups.size();
    } catch (const std::exception&) { invalid++; continue; }
    std::vector<int> rgs(n_rg);
    for (int g = 0; g < n_rg; g++) rgs[g] = g;
    for (int c = 0; c < n_cols; c++)
      for (int order = 0; order < 2; order++) {
        void* h = nullptr;
        int rc = pqemu_read_column(path.c_str(), rgs.data(), n_rg, c, order, &h);
        if (rc == 0) { ok++; pqemu_free(h); } else if (rc == 3) unsupported++; else invalid++;
      }
  }
  printf("ok=%d unsupported=%d invalid=%d\n", ok, unsupported, invalid);
  return 0;
}
