// ipc_meta_main.cpp -- stand-alone driver of the host half of the Arrow IPC scan (polars_amd/csrc/ipc_reader.hpp), built with
// -fsanitize=address,undefined by the tests: every file given (well-formed and deliberately corrupted ones) is opened, every column
// typed and located in every record batch, dictionaries and string buffers decoded.  Errors are exceptions; any out-of-bounds access
// aborts the process.  (TEST INFRASTRUCTURE: never linked into libpolars_amd.so.)
#include <cstdio>

#include "../../polars_amd/csrc/ipc_reader.hpp"

using namespace plx::ipc;

int main(int argc, char** argv) {
  int ok = 0, unsupported = 0, invalid = 0;
  long strings = 0;
  for (int a = 1; a < argc; a++) {
    try {
      std::unique_ptr<File> f = open_file(argv[a]);
      load_dictionaries(*f);
      for (size_t c = 0; c < f->footer.fields.size(); c++) {
        const Field& fl = f->footer.fields[c];
        const ColType ct = col_type(fl);
        for (size_t b = 0; b < f->batches.size(); b++) {
          const BatchMeta& bm = f->batches[b];
          Slot s = slot_of(*f, bm, (int)c);
          if (ct.dtype < 0) continue;
          if (ct.strings && !fl.has_dictionary) {
            std::vector<std::string> out;
            decode_strings(*f, fl, bm, f->body_off[b], s.buf, s.node, s.variadic, &out);
            strings += (long)out.size();
          } else if (s.buf + 2 <= bm.buffers.size()) {
            read_buffer(*f, bm, f->body_off[b], bm.buffers[s.buf + 1]);
          }
        }
      }
      ok++;
    } catch (const Unsupported&) { unsupported++;
    } catch (const std::exception&) { invalid++; }
  }
  printf("ok=%d unsupported=%d invalid=%d strings=%ld\n", ok, unsupported, invalid, strings);
  return 0;
}
