// Host-side helpers of the drop-in boundary (no device code): what stands between a prediction file on disk and the
// kernels when the library is driven through the reference's fit_eks_*() surface.
//
//   eks_csv_read_numeric   the numeric body of a DLC / Lightning Pose prediction CSV (reference eks/utils.py:188:
//                          pd.read_csv(path, header=[0, 1, 2], index_col=0)) parsed straight into a float64 matrix, one
//                          thread per block of lines over an mmap of the file.  The conversion is pandas' own default
//                          (its C tokenizer's `precise_xstrtod`, float_precision=None since pandas 1.2): at most 17
//                          significant digits accumulated in a double, the decimal exponent applied by ONE multiplication
//                          or division with an exactly representable power of ten - not strtod's correct rounding - so
//                          the doubles are the ones pd.read_csv returns, bit for bit (tests/test_csv_ingest.py checks
//                          that against pandas itself on the reference's recordings and on a fuzz corpus; pandas is the
//                          checker there, never the path).  Anything pandas would treat differently from "a number or
//                          a missing value" (quotes, a text field, ragged lines) makes the call return EKS_CSV_FALLBACK and
//                          the Python wrapper hands that file to pandas.
//   eks_host_model_flags   which EKS_FLAG_* a set of host parameter arrays allows (diagonal model, A = C = I, Q positive
//                          definite with a margin): one pass instead of a dozen NumPy calls in front of every call.
//   eks_host_gather_cols   a column block of a row-major host matrix into a contiguous buffer, threaded: the ensemble
//                          variances arrive (T, K, O) and a keypoint tile of them is a strided view (run_kalman_smoother's
//                          pipelined NumPy boundary, eks_amd/core.py).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/eks_hip.h"

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

const double kPow10[] = {
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,  1e16,
    1e17,  1e18,  1e19,  1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,  1e32,  1e33,
    1e34,  1e35,  1e36,  1e37,  1e38,  1e39,  1e40,  1e41,  1e42,  1e43,  1e44,  1e45,  1e46,  1e47,  1e48,  1e49,  1e50,
    1e51,  1e52,  1e53,  1e54,  1e55,  1e56,  1e57,  1e58,  1e59,  1e60,  1e61,  1e62,  1e63,  1e64,  1e65,  1e66,  1e67,
    1e68,  1e69,  1e70,  1e71,  1e72,  1e73,  1e74,  1e75,  1e76,  1e77,  1e78,  1e79,  1e80,  1e81,  1e82,  1e83,  1e84,
    1e85,  1e86,  1e87,  1e88,  1e89,  1e90,  1e91,  1e92,  1e93,  1e94,  1e95,  1e96,  1e97,  1e98,  1e99,  1e100, 1e101,
    1e102, 1e103, 1e104, 1e105, 1e106, 1e107, 1e108, 1e109, 1e110, 1e111, 1e112, 1e113, 1e114, 1e115, 1e116, 1e117, 1e118,
    1e119, 1e120, 1e121, 1e122, 1e123, 1e124, 1e125, 1e126, 1e127, 1e128, 1e129, 1e130, 1e131, 1e132, 1e133, 1e134, 1e135,
    1e136, 1e137, 1e138, 1e139, 1e140, 1e141, 1e142, 1e143, 1e144, 1e145, 1e146, 1e147, 1e148, 1e149, 1e150, 1e151, 1e152,
    1e153, 1e154, 1e155, 1e156, 1e157, 1e158, 1e159, 1e160, 1e161, 1e162, 1e163, 1e164, 1e165, 1e166, 1e167, 1e168, 1e169,
    1e170, 1e171, 1e172, 1e173, 1e174, 1e175, 1e176, 1e177, 1e178, 1e179, 1e180, 1e181, 1e182, 1e183, 1e184, 1e185, 1e186,
    1e187, 1e188, 1e189, 1e190, 1e191, 1e192, 1e193, 1e194, 1e195, 1e196, 1e197, 1e198, 1e199, 1e200, 1e201, 1e202, 1e203,
    1e204, 1e205, 1e206, 1e207, 1e208, 1e209, 1e210, 1e211, 1e212, 1e213, 1e214, 1e215, 1e216, 1e217, 1e218, 1e219, 1e220,
    1e221, 1e222, 1e223, 1e224, 1e225, 1e226, 1e227, 1e228, 1e229, 1e230, 1e231, 1e232, 1e233, 1e234, 1e235, 1e236, 1e237,
    1e238, 1e239, 1e240, 1e241, 1e242, 1e243, 1e244, 1e245, 1e246, 1e247, 1e248, 1e249, 1e250, 1e251, 1e252, 1e253, 1e254,
    1e255, 1e256, 1e257, 1e258, 1e259, 1e260, 1e261, 1e262, 1e263, 1e264, 1e265, 1e266, 1e267, 1e268, 1e269, 1e270, 1e271,
    1e272, 1e273, 1e274, 1e275, 1e276, 1e277, 1e278, 1e279, 1e280, 1e281, 1e282, 1e283, 1e284, 1e285, 1e286, 1e287, 1e288,
    1e289, 1e290, 1e291, 1e292, 1e293, 1e294, 1e295, 1e296, 1e297, 1e298, 1e299, 1e300, 1e301, 1e302, 1e303, 1e304, 1e305,
    1e306, 1e307, 1e308};

// pandas' default float conversion (pandas/_libs/src/parser/tokenizer.c: precise_xstrtod with decimal '.', sci 'E',
// no thousands separator, skip_trailing): restated.  [p, end) is one field; returns false when the field is not,
// in its entirety, a number in that grammar.  *is_int: no '.', no exponent (pandas would try int64 first).
bool pandas_to_double(const char* p, const char* end, double* out, bool* is_int) {
  while (p < end && is_space(*p)) ++p;
  bool negative = false;
  if (p < end && (*p == '-' || *p == '+')) {
    negative = *p == '-';
    ++p;
  }
  double number = 0.0;
  int exponent = 0, num_digits = 0, num_decimals = 0;
  const int max_digits = 17;
  *is_int = true;
  while (p < end && is_digit(*p)) {
    if (num_digits < max_digits) {
      number = number * 10.0 + (double)(*p - '0');
      ++num_digits;
    } else {
      ++exponent;
    }
    ++p;
  }
  if (p < end && *p == '.') {
    *is_int = false;
    ++p;
    while (num_digits < max_digits && p < end && is_digit(*p)) {
      number = number * 10.0 + (double)(*p - '0');
      ++p;
      ++num_digits;
      ++num_decimals;
    }
    if (num_digits >= max_digits)
      while (p < end && is_digit(*p)) ++p;
    exponent -= num_decimals;
  }
  if (num_digits == 0) return false;
  if (negative) number = -number;
  if (p < end && (*p == 'e' || *p == 'E')) {
    *is_int = false;
    const char* save = p;
    ++p;
    bool eneg = false;
    if (p < end && (*p == '-' || *p == '+')) {
      eneg = *p == '-';
      ++p;
    }
    int nd = 0, n = 0;
    while (nd < max_digits && p < end && is_digit(*p)) {
      if (n < 100000) n = n * 10 + (*p - '0');    // (saturates: anything beyond +-616 is decided below, no signed overflow)
      ++nd;
      ++p;
    }
    if (nd == 0) {
      p = save;                                   // "1e" / "1e+": the exponent marker is not consumed
    } else {
      exponent += eneg ? -n : n;
    }
  }
  if (exponent > 308) {
    return false;                                 // (ERANGE upstream: the column turns into text there - fall back)
  } else if (exponent > 0) {
    number *= kPow10[exponent];
  } else if (exponent < -308) {
    if (exponent < -616) {
      number = 0.0;
    } else {
      number /= kPow10[-308 - exponent];
      number /= kPow10[308];
    }
  } else {
    number /= kPow10[-exponent];
  }
  if (number == HUGE_VAL || number == -HUGE_VAL) return false;
  while (p < end && is_space(*p)) ++p;
  if (p != end) return false;
  *out = number;
  return true;
}

// The common case, -?digits[.digits] ending at `,` or the line's end, in one scan (no search for the field's end
// first): the same arithmetic - the first 15 digits are accumulated in an integer, which is what the double
// accumulation holds exactly at that point, the rest as pandas_to_double does.  Returns the position after the field's
// last character, or nullptr when the field is anything else (the general routine then decides).
inline const char* fast_field(const char* p, const char* lt, double* out, bool* is_int) {
  const char* q = p;
  bool negative = false;
  if (q < lt && *q == '-') {
    negative = true;
    ++q;
  }
  uint64_t acc = 0;
  int num_digits = 0, exponent = 0, num_decimals = 0;
  double number = 0.0;
  const char* d0 = q;
  while (q < lt && is_digit(*q)) {
    if (num_digits < 15) {
      acc = acc * 10u + (uint64_t)(*q - '0');
      if (++num_digits == 15) number = (double)acc;
    } else if (num_digits < 17) {
      number = number * 10.0 + (double)(*q - '0');
      ++num_digits;
    } else {
      ++exponent;
    }
    ++q;
  }
  if (q == d0) return nullptr;
  *is_int = true;
  if (q < lt && *q == '.') {
    *is_int = false;
    ++q;
    const char* f0 = q;
    while (num_digits < 17 && q < lt && is_digit(*q)) {
      if (num_digits < 15) {
        acc = acc * 10u + (uint64_t)(*q - '0');
        if (++num_digits == 15) number = (double)acc;
      } else {
        number = number * 10.0 + (double)(*q - '0');
        ++num_digits;
      }
      ++q;
      ++num_decimals;
    }
    if (num_digits >= 17)
      while (q < lt && is_digit(*q)) ++q;
    (void)f0;
    exponent -= num_decimals;
  }
  if (q != lt && *q != ',') return nullptr;        // an exponent, blanks, text: the general routine
  if (num_digits < 15) number = (double)acc;
  if (negative) number = -number;
  if (exponent > 308) return nullptr;
  if (exponent > 0) number *= kPow10[exponent];
  else if (exponent < -308) return nullptr;
  else number /= kPow10[-exponent];
  if (number == HUGE_VAL || number == -HUGE_VAL) return nullptr;
  *out = number;
  return q;
}

// the strings pandas.read_csv reads as missing by default (pandas/_libs/parsers.pyx: STR_NA_VALUES), besides the empty field
bool is_na_token(const char* p, const char* end) {
  static const char* const kNa[] = {"#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                    "<NA>", "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
  const size_t n = (size_t)(end - p);
  for (const char* s : kNa)
    if (strlen(s) == n && memcmp(s, p, n) == 0) return true;
  return false;
}
// [+-]inf / infinity, any case (the tokenizer's to_double falls back to these)
bool is_inf_token(const char* p, const char* end, double* out) {
  bool neg = false;
  if (p < end && (*p == '-' || *p == '+')) {
    neg = *p == '-';
    ++p;
  }
  const size_t n = (size_t)(end - p);
  auto eq = [&](const char* s) {
    if (strlen(s) != n) return false;
    for (size_t i = 0; i < n; ++i) {
      char c = p[i];
      if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
      if (c != s[i]) return false;
    }
    return true;
  };
  if (eq("inf") || eq("infinity")) {
    *out = neg ? -HUGE_VAL : HUGE_VAL;
    return true;
  }
  return false;
}

struct Mapped {
  const char* data = nullptr;
  size_t size = 0;
  int fd = -1;
  ~Mapped() {
    if (data && size) munmap(const_cast<char*>(data), size);
    if (fd >= 0) close(fd);
  }
};

}  // namespace

extern "C" int eks_csv_read_numeric(const char* path, int32_t skip_lines, double* out, int64_t capacity,
                                    int64_t* n_rows_out, int32_t* n_cols_out, uint8_t* col_is_int, int32_t col_capacity,
                                    int32_t n_threads) {
  if (!path || !n_rows_out || !n_cols_out) return EKS_ERR_NULL;
  Mapped m;
  m.fd = open(path, O_RDONLY);
  if (m.fd < 0) return EKS_CSV_IO;
  struct stat st;
  if (fstat(m.fd, &st) != 0) return EKS_CSV_IO;
  m.size = (size_t)st.st_size;
  if (m.size == 0) {
    *n_rows_out = 0;
    *n_cols_out = 0;
    return EKS_OK;
  }
  // (MAP_POPULATE: the page table is filled in one go - threads faulting the pages in one by one serialise on the
  //  process's mapping lock and the parse stops scaling)
  void* a = mmap(nullptr, m.size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, m.fd, 0);
  if (a == MAP_FAILED) {
    m.size = 0;
    return EKS_CSV_IO;
  }
  m.data = static_cast<const char*>(a);
  const char* p = m.data;
  const char* const end = m.data + m.size;
  for (int i = 0; i < skip_lines && p < end; ++i) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    p = nl ? nl + 1 : end;
  }
  const char* const body = p;
  if (memchr(body, '"', (size_t)(end - body)) != nullptr) return EKS_CSV_FALLBACK;      // quoting: pandas' business
  // columns: from the first non-blank body line
  const char* q = body;
  int n_cols = 0;
  while (q < end) {
    const char* nl = static_cast<const char*>(memchr(q, '\n', (size_t)(end - q)));
    const char* le = nl ? nl : end;
    const char* lt = le;
    if (lt > q && lt[-1] == '\r') --lt;
    if (lt > q) {
      n_cols = 1;
      for (const char* c = q; c < lt; ++c) n_cols += *c == ',';
      break;
    }
    q = nl ? nl + 1 : end;
  }
  *n_cols_out = n_cols;
  if (n_cols == 0) {
    *n_rows_out = 0;
    return EKS_OK;
  }
  if (n_threads < 1) n_threads = 1;
  const size_t nbytes = (size_t)(end - body);
  if (nbytes < (1u << 20)) n_threads = 1;
  // byte ranges aligned to line starts
  std::vector<const char*> cut((size_t)n_threads + 1);
  cut[0] = body;
  cut[(size_t)n_threads] = end;
  for (int t = 1; t < n_threads; ++t) {
    const char* g = body + nbytes * (size_t)t / (size_t)n_threads;
    if (g < cut[(size_t)t - 1]) g = cut[(size_t)t - 1];
    const char* nl = g < end ? static_cast<const char*>(memchr(g, '\n', (size_t)(end - g))) : nullptr;
    cut[(size_t)t] = nl ? nl + 1 : end;
  }
  // pass 1: non-blank lines per range
  std::vector<int64_t> count((size_t)n_threads, 0);
  auto count_lines = [&](int t) {
    int64_t n = 0;
    const char* c = cut[(size_t)t];
    const char* e = cut[(size_t)t + 1];
    while (c < e) {
      const char* nl = static_cast<const char*>(memchr(c, '\n', (size_t)(e - c)));
      const char* le = nl ? nl : e;
      const char* lt = le;
      if (lt > c && lt[-1] == '\r') --lt;
      n += lt > c;
      c = nl ? nl + 1 : e;
    }
    count[(size_t)t] = n;
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(count_lines, t);
    count_lines(0);
    for (auto& x : th) x.join();
  }
  int64_t n_rows = 0;
  std::vector<int64_t> first((size_t)n_threads);
  for (int t = 0; t < n_threads; ++t) {
    first[(size_t)t] = n_rows;
    n_rows += count[(size_t)t];
  }
  *n_rows_out = n_rows;
  if (!out) return EKS_OK;                                   // size query
  if (capacity < n_rows * (int64_t)n_cols) return EKS_ERR_WORKSPACE;
  if (col_is_int && col_capacity < n_cols) return EKS_ERR_WORKSPACE;
  // pass 2: parse
#ifdef MADV_POPULATE_WRITE
  {                                                          // (the output's pages too: same lock)
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(out) + 4095) & ~(uintptr_t)4095;
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(out) + (size_t)n_rows * n_cols * sizeof(double)) & ~(uintptr_t)4095;
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_POPULATE_WRITE);
  }
#endif
  std::vector<int> status((size_t)n_threads, EKS_OK);
  std::vector<std::vector<uint8_t>> ints((size_t)n_threads, std::vector<uint8_t>((size_t)n_cols, 1));
  auto parse = [&](int t) {
    const char* c = cut[(size_t)t];
    const char* e = cut[(size_t)t + 1];
    double* row = out + first[(size_t)t] * (int64_t)n_cols;
    uint8_t* isint = ints[(size_t)t].data();
    const double kNaN = std::nan("");
    while (c < e) {
      const char* nl = static_cast<const char*>(memchr(c, '\n', (size_t)(e - c)));
      const char* le = nl ? nl : e;
      const char* lt = le;
      if (lt > c && lt[-1] == '\r') --lt;
      if (lt > c) {
        const char* f = c;
        int col = 0;
        while (true) {
          if (col >= n_cols) {
            status[(size_t)t] = EKS_CSV_FALLBACK;
            return;
          }
          double v;
          bool as_int;
          const char* fe = fast_field(f, lt, &v, &as_int);
          if (fe != nullptr) {
            if (!as_int) {
              if (isint[col]) isint[col] = 0;
            } else if (std::fabs(v) >= 9007199254740992.0) {
              status[(size_t)t] = EKS_CSV_FALLBACK;
              return;
            }
            row[col++] = v;
            if (fe == lt) break;
            f = fe + 1;
            continue;
          }
          fe = static_cast<const char*>(memchr(f, ',', (size_t)(lt - f)));
          if (!fe) fe = lt;
          if (fe == f) {
            v = kNaN;                              // empty field: missing
            if (isint[col]) isint[col] = 0;        // (written once: the threads' flag arrays share cache lines)
          } else if (pandas_to_double(f, fe, &v, &as_int)) {
            if (!as_int && isint[col]) isint[col] = 0;
            else if (std::fabs(v) >= 9007199254740992.0) {
              status[(size_t)t] = EKS_CSV_FALLBACK;  // an integer a double does not hold: pandas keeps it in int64
              return;
            }
          } else if (is_na_token(f, fe)) {
            v = kNaN;
            if (isint[col]) isint[col] = 0;
          } else if (is_inf_token(f, fe, &v)) {
            if (isint[col]) isint[col] = 0;
          } else {
            status[(size_t)t] = EKS_CSV_FALLBACK;  // text (or a number pandas' grammar does not take)
            return;
          }
          row[col++] = v;
          if (fe == lt) break;
          f = fe + 1;
        }
        if (col != n_cols) {
          status[(size_t)t] = EKS_CSV_FALLBACK;    // ragged line
          return;
        }
        row += n_cols;
      }
      c = nl ? nl + 1 : e;
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(parse, t);
    parse(0);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < n_threads; ++t)
    if (status[(size_t)t] != EKS_OK) return status[(size_t)t];
  if (col_is_int) {
    for (int cidx = 0; cidx < n_cols; ++cidx) {
      uint8_t v = 1;
      for (int t = 0; t < n_threads; ++t) v &= ints[(size_t)t][(size_t)cidx] | (count[(size_t)t] == 0);
      col_is_int[cidx] = v;
    }
  }
  return EKS_OK;
}

extern "C" int eks_host_gather_cols(const void* src, int64_t n_rows, int64_t src_row_bytes, int64_t col_offset_bytes,
                                    int64_t width_bytes, void* dst, int32_t n_threads) {
  if (!src || !dst) return EKS_ERR_NULL;
  if (n_rows < 0 || width_bytes < 0 || col_offset_bytes < 0 || col_offset_bytes + width_bytes > src_row_bytes)
    return EKS_ERR_SHAPE;
  if (n_threads < 1) n_threads = 1;
  if (n_rows * width_bytes < (1 << 20)) n_threads = 1;
  const char* s = static_cast<const char*>(src) + col_offset_bytes;
  char* d = static_cast<char*>(dst);
  auto work = [&](int t) {
    const int64_t r0 = n_rows * t / n_threads, r1 = n_rows * (t + 1) / n_threads;
    for (int64_t r = r0; r < r1; ++r) memcpy(d + r * width_bytes, s + r * src_row_bytes, (size_t)width_bytes);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return EKS_OK;
}

// eks_host_model_flags: what the Python wrapper's model_flags decided with a dozen NumPy calls (0.08 ms in front of
// every call's first launch at 256 keypoints), as one pass over the host copies of the parameters.
extern "C" int eks_host_model_flags(int32_t K, int32_t D, int32_t O, const double* S0, const double* A, const double* C,
                                    const double* Q, double min_eig_ratio) {
  if (!S0 || !A || !C || !Q) return EKS_ERR_NULL;
  if (K < 0 || D < 1 || O < 1) return EKS_ERR_SHAPE;
  const int64_t dd = (int64_t)D * D;
  bool q_finite = true, q_diag = true;
  for (int64_t i = 0; i < (int64_t)K * dd && q_finite; ++i) q_finite = std::isfinite(Q[i]);
  auto is_diag = [&](const double* M, int rows) {              // (a NaN off the diagonal is not a zero)
    for (int64_t k = 0; k < K; ++k)
      for (int r = 0; r < rows; ++r)
        for (int c = 0; c < D; ++c)
          if (r != c && !(M[(k * rows + r) * D + c] == 0.0)) return false;
    return true;
  };
  q_diag = is_diag(Q, D);
  uint32_t pd = 0;
  if (q_finite) {
    if (!q_diag) return EKS_ERR_UNSUPPORTED;                   // (Q's eigenvalues decide: the caller asks LAPACK)
    bool ok = true;
    for (int64_t k = 0; k < K && ok; ++k) {
      double lo = Q[k * dd], hi = Q[k * dd];
      for (int d = 1; d < D; ++d) {
        const double v = Q[k * dd + (int64_t)d * (D + 1)];
        lo = std::min(lo, v);
        hi = std::max(hi, v);
      }
      ok = lo > min_eig_ratio * std::max(hi, 1e-300);
    }
    pd = ok ? EKS_FLAG_Q_PD : 0u;
  }
  if (D != O || !q_diag || !is_diag(S0, D) || !is_diag(A, D) || !is_diag(C, O)) return (int)pd;
  uint32_t flags = EKS_FLAG_DIAG_MODEL | pd;
  bool unit = true;
  for (int64_t k = 0; k < K && unit; ++k)
    for (int d = 0; d < D; ++d) {
      const int64_t o = k * dd + (int64_t)d * (D + 1);
      if (!(A[o] == 1.0) || !(C[o] == 1.0)) { unit = false; break; }
    }
  if (unit) flags |= EKS_FLAG_UNIT_AC;
  return (int)flags;
}

// ---------------------------------------------------------------------------------------------------------------
// eks_csv_write_table: `DataFrame.to_csv` of a float64 table with an integer index (the result tables of fit_eks_*:
// reference eks/singlecam_smoother.py:98-99, eks/multicam_smoother.py:151-152, :270-275), byte for byte.  pandas writes
// every number as Python's repr: the SHORTEST decimal string that reads back as the same double (closest to it among
// those), fixed notation for decimal exponents -4 .. 15, otherwise d.ddde+XX with a two-digit exponent; a missing
// value is the empty field.  The digits are found with the C library's correctly rounded printf / strtod: the smallest
// precision whose correctly rounded decimal reads back is the shortest one (for a power of two, whose rounding interval is
// lopsided, the neighbouring decimals are tried as well).  Row blocks are formatted concurrently and written in order.
// tests/test_csv_ingest.py holds the text to pandas' and the number formatting to Python's repr on millions of doubles.
// ---------------------------------------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <string>

namespace {

// digits (no leading / trailing zeros beyond the first) and decimal point position of the shortest repr: value =
// 0.d1d2... x 10^decpt.  Returns the number of digits.
// the exact search: the smallest precision whose correctly rounded decimal (the C library's printf) reads back
int shortest_digits_exact(double v, char* digits, int* decpt) {
  char buf[40];
  auto probe = [&](int prec) {
    snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
    return strtod(buf, nullptr) == v;
  };
  int prec;
  if (probe(15)) {
    int lo = 1, hi = 15;                          // (monotone in the precision)
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (probe(mid)) hi = mid; else lo = mid + 1;
    }
    prec = lo;
  } else {
    prec = probe(16) ? 16 : 17;
  }
  snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
  // a power of two: one decimal SHORTER may still lie in the (wider) upper half of the rounding interval
  {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    if ((bits & 0xFFFFFFFFFFFFFull) == 0 && prec > 1) {
      char b2[40];
      snprintf(b2, sizeof b2, "%.*e", prec - 2, v);          // correctly rounded, one digit fewer
      char* e = strchr(b2, 'e');
      std::string mant(b2, (size_t)(e - b2));
      std::string ex(e);
      int i = (int)mant.size() - 1;                // its upper neighbour in the last place
      while (i >= 0) {
        if (mant[(size_t)i] == '.') { --i; continue; }
        if (mant[(size_t)i] == '9') { mant[(size_t)i] = '0'; --i; continue; }
        ++mant[(size_t)i];
        break;
      }
      if (i >= 0) {
        const std::string cand = mant + ex;
        if (strtod(cand.c_str(), nullptr) == v) snprintf(buf, sizeof buf, "%s", cand.c_str());
      }
    }
  }
  int n = 0;
  const char* p = buf;
  for (; *p && *p != 'e'; ++p)
    if (*p >= '0' && *p <= '9') digits[n++] = *p;
  const int ex = atoi(p + 1);
  while (n > 1 && digits[n - 1] == '0') --n;
  digits[n] = 0;
  *decpt = ex + 1;
  return n;
}

// digits (no trailing zeros) and decimal point position of the shortest repr: value = 0.d1d2... x 10^decpt.
// ONE 17-digit conversion; shorter candidates are cut from its digits (round half up) and checked by reading them back.
// Cutting rounds twice, which differs from the correctly rounded short decimal only when what is cut is exactly 5000...
// within the 17 digits - then, and for powers of two, the exact search above decides.
int shortest_digits(double v, char* digits, int* decpt) {
  uint64_t bits;
  memcpy(&bits, &v, 8);
  if ((bits & 0xFFFFFFFFFFFFFull) == 0) return shortest_digits_exact(v, digits, decpt);
  char buf[40];
  snprintf(buf, sizeof buf, "%.16e", v);           // d.dddddddddddddddde[+-]XX
  char m[18];
  m[0] = buf[0];
  memcpy(m + 1, buf + 2, 16);
  const int ex = atoi(buf + 19);
  char cand[40];
  int cn = 0, cex = 0;
  auto cut = [&](int prec, bool* ambiguous) {      // the first prec digits of m, rounded half up -> cand / cn / cex
    bool tail_zero = true;
    for (int i = prec + 1; i < 17; ++i) tail_zero = tail_zero && m[i] == '0';
    *ambiguous = prec < 17 && m[prec] == '5' && tail_zero;
    char d[18];
    memcpy(d, m, (size_t)prec);
    cex = ex;
    if (prec < 17 && m[prec] >= '5') {
      int i = prec - 1;
      while (i >= 0 && d[i] == '9') d[i--] = '0';
      if (i >= 0) {
        ++d[i];
      } else {                                    // 99..9 -> 100..0
        d[0] = '1';
        ++cex;
      }
    }
    cn = prec;
    memcpy(cand, d, (size_t)prec);
  };
  auto reads_back = [&]() {
    char t[48];
    int o = 0;
    t[o++] = cand[0];
    t[o++] = '.';
    memcpy(t + o, cand + 1, (size_t)(cn - 1));
    o += cn - 1;
    o += snprintf(t + o, 8, "e%d", cex);
    return strtod(t, nullptr) == v;
  };
  bool amb;
  auto probe = [&](int prec, bool* any_amb) {
    cut(prec, &amb);
    *any_amb = *any_amb || amb;
    return reads_back();
  };
  bool any_amb = false;
  int prec;
  if (probe(15, &any_amb)) {
    int lo = 1, hi = 15;
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (probe(mid, &any_amb)) hi = mid; else lo = mid + 1;
    }
    prec = lo;
  } else {
    prec = probe(16, &any_amb) ? 16 : 17;
  }
  if (any_amb) return shortest_digits_exact(v, digits, decpt);
  cut(prec, &amb);
  int n = cn;
  memcpy(digits, cand, (size_t)n);
  while (n > 1 && digits[n - 1] == '0') --n;
  digits[n] = 0;
  *decpt = cex + 1;
  return n;
}

// Python's repr(float) / what DataFrame.to_csv writes (na_rep = ''); returns the length
int format_repr(double v, char* out) {
  if (v != v) return 0;
  if (v == HUGE_VAL) { memcpy(out, "inf", 3); return 3; }
  if (v == -HUGE_VAL) { memcpy(out, "-inf", 4); return 4; }
  char* o = out;
  if (std::signbit(v)) {
    *o++ = '-';
    v = -v;
  }
  if (v == 0.0) {
    memcpy(o, "0.0", 3);
    return (int)(o - out) + 3;
  }
  char d[24];
  int decpt;
  const int n = shortest_digits(v, d, &decpt);
  if (decpt > -4 && decpt <= 16) {                 // fixed notation (float_repr_style 'short', format code 'r')
    if (decpt <= 0) {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -decpt; ++i) *o++ = '0';
      memcpy(o, d, (size_t)n);
      o += n;
    } else if (decpt >= n) {
      memcpy(o, d, (size_t)n);
      o += n;
      for (int i = 0; i < decpt - n; ++i) *o++ = '0';
      *o++ = '.';
      *o++ = '0';
    } else {
      memcpy(o, d, (size_t)decpt);
      o += decpt;
      *o++ = '.';
      memcpy(o, d + decpt, (size_t)(n - decpt));
      o += n - decpt;
    }
  } else {
    *o++ = d[0];
    if (n > 1) {
      *o++ = '.';
      memcpy(o, d + 1, (size_t)(n - 1));
      o += n - 1;
    }
    *o++ = 'e';
    int ex = decpt - 1;
    *o++ = ex < 0 ? '-' : '+';
    if (ex < 0) ex = -ex;
    o += snprintf(o, 8, "%02d", ex);
  }
  return (int)(o - out);
}

}  // namespace

extern "C" int eks_format_repr(const double* values, int64_t n, char* out, int64_t capacity, int64_t* offsets) {
  // test hook: the text of n doubles, concatenated; offsets[i] .. offsets[i + 1] is number i
  if (!values || !out || !offsets) return EKS_ERR_NULL;
  int64_t at = 0;
  for (int64_t i = 0; i < n; ++i) {
    offsets[i] = at;
    if (at + 32 > capacity) return EKS_ERR_WORKSPACE;
    at += format_repr(values[i], out + at);
  }
  offsets[n] = at;
  return EKS_OK;
}

extern "C" int eks_csv_write_table(const char* path, const char* header, int64_t header_bytes, const int64_t* index,
                                   const double* values, int64_t n_rows, int32_t n_cols, int32_t n_threads) {
  if (!path || !values || (n_rows > 0 && !index) || (header_bytes > 0 && !header)) return EKS_ERR_NULL;
  if (n_rows < 0 || n_cols <= 0) return EKS_ERR_SHAPE;
  if (n_threads < 1) n_threads = 1;
  if (n_rows * (int64_t)n_cols < 20000) n_threads = 1;
  FILE* f = fopen(path, "wb");
  if (!f) return EKS_CSV_IO;
  bool ok = header_bytes == 0 || fwrite(header, 1, (size_t)header_bytes, f) == (size_t)header_bytes;
  // row blocks: formatted concurrently n_threads at a time, written in order (bounded memory: ~26 bytes per number)
  const int64_t rows_per_block = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t)(4 << 20) / (26 * (int64_t)n_cols) + 1));
  std::vector<std::string> text((size_t)n_threads);
  for (int64_t r0 = 0; ok && r0 < n_rows; r0 += rows_per_block * n_threads) {
    auto work = [&](int t) {
      const int64_t a = r0 + (int64_t)t * rows_per_block, b = std::min(n_rows, a + rows_per_block);
      std::string& s = text[(size_t)t];
      s.clear();
      if (a >= b) return;
      s.reserve((size_t)((b - a) * (26 * (int64_t)n_cols + 24)));
      char num[48];
      for (int64_t r = a; r < b; ++r) {
        s.append(num, (size_t)snprintf(num, sizeof num, "%lld", (long long)index[r]));
        const double* row = values + r * (int64_t)n_cols;
        for (int c = 0; c < n_cols; ++c) {
          s.push_back(',');
          s.append(num, (size_t)format_repr(row[c], num));
        }
        s.push_back('\n');
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; ok && t < n_threads; ++t)
      ok = text[(size_t)t].empty() || fwrite(text[(size_t)t].data(), 1, text[(size_t)t].size(), f) == text[(size_t)t].size();
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? EKS_OK : EKS_CSV_IO;
}

// How much faster do n_threads threads of this process get a fixed amount of independent work done than one?
// (A sandbox may give a process several CPUs and still run its threads one at a time; the writer above only pays
// when they run side by side - its per-number cost on ONE thread is several times Python's own repr.)
extern "C" double eks_host_thread_speedup(int32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  auto spin = [](double* out) {
    double x = 1.0;
    for (int i = 0; i < 2000000; ++i) x = x * 1.0000001 + 1e-9;      // (~5 ms: well above the cost of starting the threads)
    *out = x;
  };
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<double> sink((size_t)n_threads * 16);
  double t0 = now();
  spin(&sink[0]);
  const double one = now() - t0;
  t0 = now();
  {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(spin, &sink[(size_t)t * 16]);
    spin(&sink[0]);
    for (auto& x : th) x.join();
  }
  const double many = now() - t0;
  return many > 0.0 ? one * n_threads / many : 1.0;
}
