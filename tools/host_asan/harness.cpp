// AddressSanitizer / UBSan harness for the library's HOST code (eks_amd/csrc/eks_host.hip: CSV body reader, table writer,
// number formatter, column gather) - built for the CPU with g++ (run.sh), no GPU involved.  Every file of the corpus
// (make_corpus.py: well-formed tables and adversarial ones - truncated, ragged, CRLF, NUL bytes, 400-digit fields, no
// trailing newline, empty) goes through the reader's two-call protocol on 1, 2 and 5 threads with exact and short
// capacities; what parses is formatted, written with the table writer and read back.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <dirent.h>
#include "../../include/eks_hip.h"

static int check_file(const std::string& path, const std::string& tmp) {
  int problems = 0;
  for (int skip : {0, 3}) {
    for (int threads : {1, 2, 5}) {
      int64_t nr = -1;
      int32_t nc = -1;
      int rc = eks_csv_read_numeric(path.c_str(), skip, nullptr, 0, &nr, &nc, nullptr, 0, threads);
      if (rc != EKS_OK) continue;                       // fallback / io: nothing more to do with this file
      if (nr < 0 || nc < 0) { std::printf("bad size answer %s\n", path.c_str()); ++problems; continue; }
      std::vector<double> body((size_t)nr * nc + 1, -7.0);
      std::vector<uint8_t> isint((size_t)nc + 1, 9);
      if (nr * nc > 0) {                                 // a capacity one short must be refused, not overrun
        rc = eks_csv_read_numeric(path.c_str(), skip, body.data(), nr * nc - 1, &nr, &nc, isint.data(), nc, threads);
        if (rc == EKS_OK) { std::printf("short capacity accepted %s\n", path.c_str()); ++problems; }
      }
      rc = eks_csv_read_numeric(path.c_str(), skip, body.data(), nr * nc, &nr, &nc, isint.data(), nc, threads);
      if (rc != EKS_OK && rc != EKS_CSV_FALLBACK) { std::printf("rc %d on the parse of %s\n", rc, path.c_str()); ++problems; continue; }
      if (body[(size_t)nr * nc] != -7.0 || isint[nc] != 9) { std::printf("guard overwritten %s\n", path.c_str()); ++problems; }
      if (rc != EKS_OK || nr == 0 || nc < 2) continue;
      // format every value; write the table (first column as the index when it is integral) and read it back
      std::vector<int64_t> offs((size_t)nr * nc + 1);
      std::vector<char> text((size_t)nr * nc * 26 + 16);
      if (eks_format_repr(body.data(), nr * nc, text.data(), (int64_t)text.size(), offs.data()) != EKS_OK) { ++problems; continue; }
      if (eks_format_repr(body.data(), nr * nc, text.data(), offs[(size_t)nr * nc] - 1, offs.data()) == EKS_OK && offs[(size_t)nr * nc] > 0) {
        std::printf("formatter accepted a short buffer %s\n", path.c_str()); ++problems;
      }
      std::vector<int64_t> index(nr);
      std::vector<double> vals((size_t)nr * (nc - 1));
      for (int64_t r = 0; r < nr; ++r) {
        const double v = body[(size_t)r * nc];
        index[r] = std::isfinite(v) && std::fabs(v) < 9e15 ? (int64_t)v : r;
        for (int c = 1; c < nc; ++c) vals[(size_t)r * (nc - 1) + c - 1] = body[(size_t)r * nc + c];
      }
      const char* header = "a,b\nc,d\ne,f\n";
      if (eks_csv_write_table(tmp.c_str(), header, (int64_t)std::strlen(header), index.data(), vals.data(), nr, nc - 1, threads) != EKS_OK) {
        std::printf("writer failed %s\n", path.c_str()); ++problems; continue;
      }
      int64_t nr2 = 0; int32_t nc2 = 0;
      rc = eks_csv_read_numeric(tmp.c_str(), 3, nullptr, 0, &nr2, &nc2, nullptr, 0, threads);
      if (rc != EKS_OK || nr2 != nr || nc2 != nc) { std::printf("read-back shape %s: rc %d %lld x %d\n", path.c_str(), rc, (long long)nr2, nc2); ++problems; continue; }
      std::vector<double> back((size_t)nr * nc);
      eks_csv_read_numeric(tmp.c_str(), 3, back.data(), nr * nc, &nr2, &nc2, nullptr, 0, threads);
      // (pandas' conversion is not correctly rounded: the read-back may differ from the original by an ulp - only NaN-ness
      //  and size are compared here; the values are compared against pandas in tests/test_csv_ingest.py)
      for (int64_t i = 0; i < nr * nc; ++i)
        if ((i % nc) != 0 && std::isnan(back[i]) != std::isnan(body[i])) { std::printf("NaN pattern changed %s\n", path.c_str()); ++problems; break; }
    }
  }
  return problems;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "corpus";
  const std::string tmp = dir + "/_written.csv";
  int files = 0, problems = 0;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() < 5 || n.substr(n.size() - 4) != ".csv" || n[0] == '_') continue;
      problems += check_file(dir + "/" + n, tmp);
      ++files;
    }
    closedir(d);
  }
  // the column gather: tiles of a row-major matrix, exact bounds
  std::vector<float> src(3000 * 80), dst(3000 * 16 + 1, -3.f);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (float)i;
  for (int threads : {1, 3, 8}) {
    if (eks_host_gather_cols(src.data(), 3000, 80 * 4, 64 * 4, 16 * 4, dst.data(), threads) != EKS_OK) ++problems;
    if (dst[3000 * 16] != -3.f || dst[16] != src[80 + 64]) { std::printf("gather wrong\n"); ++problems; }
    if (eks_host_gather_cols(src.data(), 3000, 80 * 4, 70 * 4, 16 * 4, dst.data(), threads) == EKS_OK) { std::printf("gather past the row accepted\n"); ++problems; }
  }
  std::printf("%d files, %d problems\n", files, problems);
  return problems ? 1 : 0;
}
