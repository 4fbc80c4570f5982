// spmx_encode: the command line of the reference's spm_encode (src/spm_encode_main.cc) for the one output this engine
// produces -- ids -- over the C ABI of include/spmx.h: file (or stdin) in, one line of space-separated ids per input
// line out (--output_format=id, the reference's bytes), or flat binary ids (--output_format=bin).
//   spmx_encode --model=M [--input=F] [--output=F] [--output_format=id|bin] [--extra_options=bos:eos] [--device=N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>

#include "../include/spmx.h"

int main(int argc, char **argv) {
  std::string model, input, output, format = "id", extra;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&](const char *name, std::string *dst) {
      const std::string k = std::string("--") + name + "=";
      if (a.compare(0, k.size(), k) == 0) { *dst = a.substr(k.size()); return true; }
      return false;
    };
    std::string dev;
    if (val("model", &model) || val("input", &input) || val("output", &output) || val("output_format", &format) ||
        val("extra_options", &extra)) continue;
    if (val("device", &dev)) { device = atoi(dev.c_str()); continue; }
    if (a[0] != '-' && input.empty()) { input = a; continue; }
    fprintf(stderr, "unknown argument: %s\n", a.c_str());
    return 2;
  }
  if (model.empty()) { fprintf(stderr, "usage: spmx_encode --model=M [--input=F] [--output=F] [--output_format=id|bin] [--extra_options=..]\n"); return 2; }
  spmx_handle *h = nullptr;
  if (spmx_create_from_file(model.c_str(), device, &h) != 0) { fprintf(stderr, "%s\n", spmx_last_error(nullptr)); return 1; }
  if (!extra.empty() && spmx_set_encode_extra_options(h, extra.c_str()) != 0) { fprintf(stderr, "%s\n", spmx_last_error(h)); return 1; }
  // stdin / stdout go through temporary files: the library maps its input
  std::string in_path = input, out_path = output;
  char tin[] = "/tmp/spmx_encode_in_XXXXXX", tout[] = "/tmp/spmx_encode_out_XXXXXX";
  if (in_path.empty()) {
    const int fd = mkstemp(tin);
    if (fd < 0) { perror("mkstemp"); return 1; }
    FILE *f = fdopen(fd, "wb");
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), stdin)) > 0) fwrite(buf, 1, n, f);
    fclose(f);
    in_path = tin;
  }
  if (out_path.empty()) { const int fd = mkstemp(tout); if (fd < 0) { perror("mkstemp"); return 1; } close(fd); out_path = tout; }
  uint64_t ns = 0, ni = 0;
  const int rc = spmx_encode_file(h, in_path.c_str(), out_path.c_str(), format.c_str(), &ns, &ni);
  if (rc != 0) fprintf(stderr, "%s\n", spmx_last_error(h));
  if (rc == 0 && output.empty()) {
    FILE *f = fopen(out_path.c_str(), "rb");
    char buf[1 << 16];
    size_t n;
    while (f && (n = fread(buf, 1, sizeof(buf), f)) > 0) fwrite(buf, 1, n, stdout);
    if (f) fclose(f);
  }
  if (input.empty()) remove(tin);
  if (output.empty()) remove(tout);
  spmx_destroy(h);
  if (rc == 0) fprintf(stderr, "spmx_encode: %llu sentences, %llu ids\n", static_cast<unsigned long long>(ns), static_cast<unsigned long long>(ni));
  return rc == 0 ? 0 : 1;
}
