/*
 * model_io.hpp -- multi-threaded writers for LIBSVM model files and LIBSVM data files, and the reader for well-formed model files (host code only;
 * the formats plssvm-train writes and plssvm-predict reads on either side of the hot path).
 *
 * Format (citations relative to /root/reference):
 *   - a model file is a header of "key value" lines that ends with "SV", then one line per support vector: "alpha idx:val idx:val ... " with
 *     every number as {:.10e}, zero features left out, one-based indices, a blank after EVERY token (include/plssvm/detail/io/libsvm_model_parsing.hpp:371-414;
 *     header :296-342), the support vectors grouped by class in the order of the header's "label" line (:416-499);
 *   - a data file line is "label idx:val idx:val ... " with the same number format, or without the label (include/plssvm/detail/io/libsvm_parsing.hpp:244-296);
 *   - the reader's header rules: libsvm_model_parsing.hpp:64-262 (keys in any order and case, "SV" ends the header), its body rules those of a data file
 *     whose label column holds alpha (model.hpp:187 -> libsvm_parsing.hpp:118-229).
 * The reference formats with per-thread buffers under "#pragma omp parallel" and flushes them inside a critical section in whatever order the threads arrive
 * (:422-460; "the resulting order of the data points is unspecified" within a class).  Here the rows are cut into chunks, every thread formats the next chunk
 * into its own buffer, and the chunks reach the file in ROW ORDER: a thread learns its file offset from its predecessor and then writes with pwrite while
 * the others keep formatting -- the same bytes for any number of threads.
 * The reader is a fast path like libsvm_reader.hpp: anything irregular makes it decline without a diagnosis and the caller (plssvm_amd/model.py) parses with
 * the reference-exact Python implementation, which words the error.  So the accepted language may be narrower than the format, never wider.
 */
#ifndef PLSSVM_AMD_MODEL_IO_HPP_
#define PLSSVM_AMD_MODEL_IO_HPP_

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "libsvm_reader.hpp"
#include "text_file.hpp"

namespace lssvm {

/* ------------------------------------------------------------------ writer ------------------------------------------------------------------ */

/* "{:.10e}" of fmt / "%.10e" of printf / f"{v:.10e}" of Python: the same characters for every finite double (correctly rounded, exponent of at least
 * two digits); infinities as "inf" / "-inf", every NaN as "nan" (Python's spelling; fmt writes the sign of a NaN).  At most 24 characters. */
inline char *put_sci10(char *p, double v) {
    if (std::isnan(v)) {
        std::memcpy(p, "nan", 3);
        return p + 3;
    }
    return std::to_chars(p, p + 32, v, std::chars_format::scientific, 10).ptr;
}
inline char *put_index(char *p, std::uint64_t i) { return std::to_chars(p, p + 24, i).ptr; }

/* Where the first token of a line comes from. */
template <typename T>
struct AlphaPrefix {  // model files: "{:.10e} " of alpha[row]
    const T *alpha;
    std::size_t bound() const { return 32; }
    char *put(char *p, std::size_t row) const {
        p = put_sci10(p, static_cast<double>(alpha[row]));
        *p++ = ' ';
        return p;
    }
};
struct TextPrefix {  // data files: the caller's label text of the row + ' ' (text == nullptr: no label column)
    const char *text;
    const std::uint64_t *offsets;  // row r: [offsets[r], offsets[r + 1])
    std::size_t longest;
    std::size_t bound() const { return longest + 1; }
    char *put(char *p, std::size_t row) const {
        if (text == nullptr) return p;
        const std::size_t len = static_cast<std::size_t>(offsets[row + 1] - offsets[row]);
        std::memcpy(p, text + offsets[row], len);
        p += len;
        *p++ = ' ';
        return p;
    }
};
struct IntegerPrefix {  // data files: labels that are whole numbers, written like fmt's "{}" of an integer
    const std::int64_t *labels;
    std::size_t bound() const { return 24; }
    char *put(char *p, std::size_t row) const {
        p = std::to_chars(p, p + 24, labels[row]).ptr;
        *p++ = ' ';
        return p;
    }
};

class RowWriter {
  public:
    /* creates / truncates the file and writes `header` (may be empty); false with errno kept in error() */
    bool open(const char *path, const char *header, std::size_t header_len) {
        fd_ = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        if (fd_ < 0) {
            error_ = errno;
            return false;
        }
        offset_ = 0;
        return header_len == 0 || put_at(header, header_len, 0, true);
    }

    /* rows[order[k]] for k in [0, count) (order == nullptr: rows 0 .. count-1), in that order */
    template <typename T, typename Prefix>
    bool write_rows(const T *X, std::size_t num_features, std::size_t ldx, const std::uint64_t *order, std::size_t count, const Prefix &prefix) {
        if (count == 0) return error_ == 0;
        // a chunk: about 1 MiB of text (the reference's per-thread buffer, libsvm_model_parsing.hpp:399), at least one row
        const std::size_t row_bound = prefix.bound() + num_features * kEntryBound + 2;
        const std::size_t rows_per_chunk = std::max<std::size_t>(1, (std::size_t(1) << 20) / row_bound);
        const std::size_t chunks = (count + rows_per_chunk - 1) / rows_per_chunk;
        const unsigned nt = static_cast<unsigned>(std::min<std::size_t>(io_threads(count * (num_features + 1), 1 << 16), chunks));
        std::atomic<std::size_t> next_chunk{ 0 };
        std::size_t turn = 0;  // the chunk whose offset is known (guarded by m)
        std::mutex m;
        std::condition_variable cv;
        std::atomic<int> failed{ 0 };
        io_parallel(nt, nt, [&](unsigned, std::size_t, std::size_t) {
            std::vector<char> buf(rows_per_chunk * row_bound);
            while (true) {
                const std::size_t c = next_chunk.fetch_add(1, std::memory_order_relaxed);
                if (c >= chunks) return;
                const std::size_t lo = c * rows_per_chunk, hi = std::min(count, lo + rows_per_chunk);
                char *p = buf.data();
                if (failed.load(std::memory_order_relaxed) == 0) {
                    for (std::size_t k = lo; k < hi; ++k) {
                        const std::size_t row = order != nullptr ? static_cast<std::size_t>(order[k]) : k;
                        p = prefix.put(p, row);
                        const T *x = X + row * ldx;
                        for (std::size_t j = 0; j < num_features; ++j) {
                            if (x[j] != T(0)) {  // (a NaN is != 0 and is written, like the reference's comparison)
                                p = put_index(p, j + 1);
                                *p++ = ':';
                                p = put_sci10(p, static_cast<double>(x[j]));
                                *p++ = ' ';
                            }
                        }
                        *p++ = '\n';
                    }
                }
                const std::size_t len = static_cast<std::size_t>(p - buf.data());
                std::size_t at = 0;
                {
                    std::unique_lock<std::mutex> lock(m);
                    cv.wait(lock, [&] { return turn == c; });
                    at = offset_;
                    offset_ += len;
                    ++turn;
                }
                cv.notify_all();
                if (len != 0 && !put_at(buf.data(), len, at, false)) failed.store(1, std::memory_order_relaxed);
            }
        });
        return failed.load() == 0 && error_ == 0;
    }

    bool close() {
        if (fd_ >= 0 && ::close(fd_) != 0 && error_ == 0) error_ = errno;
        fd_ = -1;
        return error_ == 0;
    }
    ~RowWriter() {
        if (fd_ >= 0) ::close(fd_);
    }
    int error() const { return error_; }
    std::uint64_t bytes() const { return offset_; }

    // "18446744073709551615:-1.2345678901e-308 " -> 20 + 1 + 18 + 1 = 40 (the reference reserves 48, libsvm_model_parsing.hpp:388-396)
    static constexpr std::size_t kEntryBound = 48;

  private:
    bool put_at(const char *data, std::size_t len, std::size_t at, bool advance) {
        std::size_t done = 0;
        while (done < len) {
            const ssize_t w = ::pwrite(fd_, data + done, len - done, static_cast<off_t>(at + done));
            if (w < 0) {
                if (errno == EINTR) continue;
                error_ = errno;
                return false;
            }
            done += static_cast<std::size_t>(w);
        }
        if (advance) offset_ += len;
        return true;
    }

    int fd_ = -1;
    std::atomic<int> error_{ 0 };
    std::size_t offset_ = 0;
};

/* ------------------------------------------------------------------ model reader ------------------------------------------------------------------ */

struct ModelHeader {
    int kernel_type = -1;  // 0 linear, 1 polynomial, 2 rbf
    bool has_degree = false, has_gamma = false, has_coef0 = false;
    long long degree = 0;
    double gamma = 0.0, coef0 = 0.0, rho = 0.0;
    std::uint64_t nr_class = 0, total_sv = 0;
    std::vector<std::string> labels;
    std::vector<std::uint64_t> nr_sv;
};

class ModelFile {
  public:
    /* false: the file cannot be read, or is anything but a plainly well-formed model file (the caller's parser decides) */
    bool open(const char *path) {
        if (!body_.text().open(path)) return false;
        return parse();
    }
    bool open_text(std::string text) {  // tests
        body_.text().adopt(std::move(text));
        return parse();
    }
    const ModelHeader &header() const { return header_; }
    std::size_t num_support_vectors() const { return body_.num_points(); }
    std::size_t num_features() const { return body_.num_features(); }

    /* the support vectors (dense row-major, ldx >= num_features) and their weights */
    template <typename T>
    bool fill(T *sv, std::size_t ldx, T *alpha) const {
        std::vector<double> a(body_.num_points());
        if (!body_.fill<T>(sv, ldx, a.data())) return false;
        for (std::size_t i = 0; i < a.size(); ++i) alpha[i] = static_cast<T>(a[i]);
        return true;
    }

  private:
    static bool key_is(const std::string &low, const char *key) { return low.compare(0, std::strlen(key), key) == 0; }

    /* a whole token as a number, in the spellings BOTH std::from_chars and Python's float() / int() accept */
    static bool to_real(const std::string &s, double &out) {
        if (s.empty() || s[0] == '+') return false;
        for (const char c : s) {
            if (!((c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.' || c == 'e')) return false;  // (the value is lower-cased; "inf" / "nan" parameters are left to Python)
        }
        const auto r = std::from_chars(s.data(), s.data() + s.size(), out);
        return r.ec == std::errc() && r.ptr == s.data() + s.size();
    }
    template <typename I>
    static bool to_integer(const std::string &s, I &out, bool may_be_negative) {
        if (s.empty()) return false;
        std::size_t i = 0;
        if (s[0] == '-') {
            if (!may_be_negative) return false;
            i = 1;
        }
        if (i >= s.size()) return false;
        for (std::size_t k = i; k < s.size(); ++k) {
            if (s[k] < '0' || s[k] > '9') return false;
        }
        const auto r = std::from_chars(s.data(), s.data() + s.size(), out);
        return r.ec == std::errc() && r.ptr == s.data() + s.size();
    }
    static std::vector<std::string> split_blanks(const std::string &s) {
        std::vector<std::string> out;
        std::size_t i = 0;
        while (i < s.size()) {
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
            std::size_t j = i;
            while (j < s.size() && s[j] != ' ' && s[j] != '\t') ++j;
            if (j > i) out.push_back(s.substr(i, j - i));
            i = j;
        }
        return out;
    }

    bool parse() {
        const TextFile &text = body_.text();
        header_ = ModelHeader{};
        bool svm_type = false, nr_class = false, total_sv = false, rho = false, seen_sv = false;
        std::size_t at = 0;
        TextFile::Line line{};
        // the header: a handful of lines (libsvm_model_parsing.hpp:100-196); every key at most once here
        for (int count = 0; count < 64 && text.next_line(at, '#', line, at); ++count) {
            std::string raw(text.data() + line.begin, line.end - line.begin);
            while (!raw.empty() && (raw.back() == ' ' || raw.back() == '\t' || raw.back() == '\v' || raw.back() == '\f')) raw.pop_back();
            for (const char c : raw) {
                if (static_cast<unsigned char>(c) >= 0x80 || (static_cast<unsigned char>(c) < 0x20 && c != '\t')) return false;
            }
            std::string low = raw;
            for (char &c : low) c = (c >= 'A' && c <= 'Z') ? static_cast<char>(c - 'A' + 'a') : c;
            if (low == "sv") {
                seen_sv = true;
                break;
            }
            const std::size_t blank = low.find(' ');
            if (blank == std::string::npos) return false;
            std::string value = low.substr(blank + 1);
            while (!value.empty() && (value.front() == ' ' || value.front() == '\t')) value.erase(value.begin());
            const std::string key = low.substr(0, blank);
            if (key == "svm_type") {
                if (svm_type || value != "c_svc") return false;
                svm_type = true;
            } else if (key == "kernel_type") {
                if (header_.kernel_type != -1) return false;
                if (value == "linear") header_.kernel_type = 0;
                else if (value == "polynomial") header_.kernel_type = 1;
                else if (value == "rbf") header_.kernel_type = 2;
                else return false;  // (the reference also knows "poly" and numbers: left to the Python parser)
            } else if (key == "gamma") {
                if (header_.has_gamma || !to_real(value, header_.gamma)) return false;
                header_.has_gamma = true;
            } else if (key == "degree") {
                if (header_.has_degree || !to_integer(value, header_.degree, true)) return false;
                header_.has_degree = true;
            } else if (key == "coef0") {
                if (header_.has_coef0 || !to_real(value, header_.coef0)) return false;
                header_.has_coef0 = true;
            } else if (key == "nr_class") {
                if (nr_class || !to_integer(value, header_.nr_class, false)) return false;
                nr_class = true;
            } else if (key == "total_sv") {
                if (total_sv || !to_integer(value, header_.total_sv, false) || header_.total_sv == 0) return false;
                total_sv = true;
            } else if (key == "rho") {
                if (rho || !to_real(value, header_.rho)) return false;
                rho = true;
            } else if (key == "label") {
                if (!header_.labels.empty()) return false;
                header_.labels = split_blanks(raw.substr(blank + 1));  // the labels keep their case
                if (header_.labels.size() < 2) return false;
                std::vector<std::string> sorted = header_.labels;
                std::sort(sorted.begin(), sorted.end());
                if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return false;
            } else if (key == "nr_sv") {
                if (!header_.nr_sv.empty()) return false;
                for (const std::string &tok : split_blanks(value)) {
                    std::uint64_t v = 0;
                    if (!to_integer(tok, v, false)) return false;
                    header_.nr_sv.push_back(v);
                }
                if (header_.nr_sv.size() < 2) return false;
            } else {
                return false;
            }
        }
        if (!seen_sv || !svm_type || header_.kernel_type == -1 || !nr_class || !total_sv || !rho || header_.labels.empty() || header_.nr_sv.empty()) return false;
        if (header_.kernel_type == 0 && (header_.has_degree || header_.has_gamma || header_.has_coef0)) return false;
        if (header_.kernel_type == 2 && (header_.has_degree || header_.has_coef0)) return false;
        if (header_.nr_class != header_.labels.size() || header_.nr_class != header_.nr_sv.size()) return false;
        std::uint64_t sum = 0;
        for (const std::uint64_t v : header_.nr_sv) {
            if (v > header_.total_sv) return false;
            sum += v;
        }
        if (sum != header_.total_sv) return false;
        // the body: a LIBSVM data file whose label column is alpha, every line labelled
        body_.index(at, 0);
        if (body_.num_points() != header_.total_sv) return false;
        return body_.scan() && body_.has_label();
    }

    LibsvmFile body_;
    ModelHeader header_;
};

}  // namespace lssvm

#endif  // PLSSVM_AMD_MODEL_IO_HPP_
