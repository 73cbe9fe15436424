/*
 * arff_reader.hpp -- multi-threaded reader for well-formed ARFF data files (host code only), the second data format of
 * plssvm::data_set beside LIBSVM (libsvm_reader.hpp is its sibling).
 *
 * The format rules are the reference's (citations relative to /root/reference):
 *   - lines end at '\r' or '\n', are left-trimmed, and are dropped when empty or starting with '%' (src/plssvm/detail/io/file_reader.cpp:179-205);
 *   - header (include/plssvm/detail/io/arff_parsing.hpp:57-170): "@RELATION name" first, "@ATTRIBUTE name NUMERIC" per feature, at most one nominal
 *     attribute with the reserved name "class" that lists the labels ("@ATTRIBUTE class {-1,1}"), keywords in any case, "@DATA" ends it;
 *   - data (arff_parsing.hpp:196-372): dense rows "v,v,...,label" with one value per attribute in header order, or sparse rows "{index value,index value}"
 *     with zero-based attribute indices (missing features are zeros, the label must be given); every label must be one the header lists.
 * This reader is the FAST PATH for files that follow those rules to the letter.  Anything else -- a token that does not convert as a whole, a quoted or
 * blank-holding name, a header line it does not know, a label type other than a number -- makes it report failure without a diagnosis; the caller
 * (plssvm_amd/io_arff.py) then re-parses with the line-by-line Python implementation, which raises the reference's exact error messages.  So the
 * accepted language here may be narrower than the format, never wider.
 */
#ifndef PLSSVM_AMD_ARFF_READER_HPP_
#define PLSSVM_AMD_ARFF_READER_HPP_

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

namespace lssvm {

class ArffFile {
  public:
    /* reads the file and indexes its lines; returns false if the file cannot be read */
    bool open(const char *path) {
        std::FILE *f = std::fopen(path, "rb");
        if (f == nullptr) return false;
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (size < 0) {
            std::fclose(f);
            return false;
        }
        text_.resize(static_cast<std::size_t>(size));
        const std::size_t got = size > 0 ? std::fread(&text_[0], 1, text_.size(), f) : 0;
        std::fclose(f);
        if (got != text_.size()) return false;
        const char *b = text_.data();
        const char *e = b + text_.size();
        for (const char *p = b; p < e;) {
            const char *q = p;
            while (q < e && *q != '\n' && *q != '\r') ++q;
            const char *s = p;
            while (s < q && is_space(*s)) ++s;
            if (s < q && *s != '%') lines_.push_back({ static_cast<std::size_t>(s - b), static_cast<std::size_t>(q - b) });
            p = q + 1;
        }
        return true;
    }

    /* the header (sequential: a handful of lines) and a validating pass over the data rows.  int_labels: the caller's label type is an integer -- the class
     * labels must then be written as plain integers (the reference converts the longest integer PREFIX of a token, this reader only whole tokens). */
    bool scan(bool int_labels) {
        int_labels_ = int_labels;
        std::size_t i = 0;
        num_features_ = 0;
        has_label_ = false;
        for (; i < lines_.size(); ++i) {
            const char *p = text_.data() + lines_[i].begin;
            const char *e = line_end(i);
            if (*p != '@') return false;  // (the reference skips such lines in the header: left to it)
            if (keyword(p, e, "@RELATION")) {
                if (i != 0) return false;
                const char *n = skip_spaces(p + 9, e);
                if (n == p + 9 || !plain_name(n, e)) return false;
            } else if (keyword(p, e, "@ATTRIBUTE")) {
                const char *n = skip_spaces(p + 10, e);
                if (n == p + 10) return false;
                const char *ne = n;
                while (ne < e && !is_space(*ne)) ++ne;
                if (!plain_name(n, ne)) return false;
                const char *t = skip_spaces(ne, e);
                if (t == ne) return false;
                if (equals_nocase(n, ne, "CLASS")) {
                    if (has_label_ || *t != '{' || e[-1] != '}') return false;
                    if (!class_labels(t + 1, e - 1)) return false;
                    has_label_ = true;
                    label_idx_ = num_features_;
                } else {
                    if (!equals_nocase(t, e, "NUMERIC")) return false;
                    ++num_features_;
                }
            } else if (keyword(p, e, "@DATA")) {
                if (e != p + 5) return false;
                break;
            } else {
                return false;
            }
        }
        if (num_features_ == 0 || i + 1 >= lines_.size()) return false;  // (no features, no @DATA, or no data rows: the reference's errors)
        first_row_ = i + 1;
        // validating pass (every row is walked once here and once in fill: the walk is cheaper than a second copy of the values)
        std::atomic<bool> ok{ true };
        run_parallel(num_threads(), [&](std::size_t lo, std::size_t hi) {
            for (std::size_t r = lo; r < hi && ok.load(std::memory_order_relaxed); ++r) {
                double label = 0.0;
                if (!walk_row(r, [](std::size_t, double) {}, label)) {
                    ok.store(false, std::memory_order_relaxed);
                    return;
                }
            }
        });
        return ok.load();
    }

    /* dense row-major matrix (leading dimension ldx >= num_features, zeroed here) and the labels (always double) */
    template <typename T>
    bool fill(T *X, std::size_t ldx, double *labels) const {
        if (ldx < num_features_) return false;
        std::atomic<bool> ok{ true };
        run_parallel(num_threads(), [&](std::size_t lo, std::size_t hi) {
            for (std::size_t r = lo; r < hi; ++r) {
                T *row = X + r * ldx;
                std::fill(row, row + ldx, T(0));
                double label = 0.0;
                if (!walk_row(r, [&](std::size_t feature, double v) { row[feature] = static_cast<T>(v); }, label)) {
                    ok.store(false, std::memory_order_relaxed);
                    return;
                }
                if (labels != nullptr && has_label_) labels[r] = label;
            }
        });
        return ok.load();
    }

    std::size_t num_points() const { return lines_.size() - first_row_; }
    std::size_t num_features() const { return num_features_; }
    bool has_label() const { return has_label_; }

  private:
    struct Line {
        std::size_t begin, end;
    };

    static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }
    static const char *skip_spaces(const char *p, const char *e) {
        while (p < e && is_space(*p)) ++p;
        return p;
    }
    /* (lines are LEFT-trimmed only, as the reference's are: blanks at the end of a header line or behind a closing brace change what the reference reads, so they
     * are not trimmed away here -- such lines simply fail the checks below and go to the Python parser) */
    const char *line_end(std::size_t i) const { return text_.data() + lines_[i].end; }
    static char upper(char c) { return (c >= 'a' && c <= 'z') ? static_cast<char>(c - 'a' + 'A') : c; }
    static bool equals_nocase(const char *p, const char *e, const char *word) {
        for (; *word != '\0'; ++word, ++p) {
            if (p >= e || upper(*p) != *word) return false;
        }
        return p == e;
    }
    /* the line starts with `word` (any case) followed by a blank or the end of the line */
    static bool keyword(const char *p, const char *e, const char *word) {
        for (; *word != '\0'; ++word, ++p) {
            if (p >= e || upper(*p) != *word) return false;
        }
        return p == e || is_space(*p);
    }
    /* a name without blanks, quotes or braces (quoted names and their rules are the Python parser's) */
    static bool plain_name(const char *p, const char *e) {
        if (p >= e) return false;
        for (; p < e; ++p) {
            if (is_space(*p) || *p == '"' || *p == '\'' || *p == '{' || *p == '}' || *p == ',') return false;
        }
        return true;
    }

    /* a whole token as a finite number ("+1" is left to the reference's conversion, like everything std::from_chars does not take) */
    bool number(const char *p, const char *e, double &v, bool as_label) const {
        p = skip_spaces(p, e);
        while (e > p && is_space(e[-1])) --e;
        if (p >= e) return false;
        if (as_label && int_labels_) {
            long long iv = 0;
            const auto r = std::from_chars(p, e, iv);
            if (r.ec != std::errc() || r.ptr != e) return false;
            v = static_cast<double>(iv);
            return true;
        }
        const auto r = std::from_chars(p, e, v);
        return r.ec == std::errc() && r.ptr == e && std::isfinite(v);
    }

    bool class_labels(const char *p, const char *e) {
        allowed_.clear();
        while (true) {
            const char *c = p;
            while (c < e && *c != ',') ++c;
            double v = 0.0;
            if (!number(p, c, v, true)) return false;
            allowed_.push_back(v);
            if (c >= e) break;
            p = c + 1;
        }
        std::sort(allowed_.begin(), allowed_.end());
        return allowed_.size() >= 2 && std::adjacent_find(allowed_.begin(), allowed_.end()) == allowed_.end();
    }
    bool allowed(double label) const { return std::binary_search(allowed_.begin(), allowed_.end(), label); }

    /* walks data row r: `emit(feature index, value)` per given feature; the label (if the file has one).  false = not well formed. */
    template <typename Emit>
    bool walk_row(std::size_t r, Emit &&emit, double &label) const {
        const std::size_t li = first_row_ + r;
        const char *p = text_.data() + lines_[li].begin;
        const char *e = line_end(li);
        const std::size_t num_attributes = num_features_ + (has_label_ ? 1 : 0);
        if (*p == '@') return false;
        if (*p == '{') {  // sparse: {index value,index value,...}
            if (e[-1] != '}' || e - p < 2) return false;
            ++p;
            --e;
            bool class_set = false;
            if (p >= e) return !has_label_;  // "{}": all zeros (without a class attribute only)
            while (true) {
                std::size_t index = 0;
                const auto ri = std::from_chars(p, e, index);  // (digits right behind the brace / the blanks after a comma; a SPACE ends the index: the reference looks for ' ')
                if (ri.ec != std::errc() || ri.ptr >= e || *ri.ptr != ' ' || index >= num_attributes) return false;
                const char *c = ri.ptr;
                while (c < e && *c != ',') ++c;
                double v = 0.0;
                const bool is_label = has_label_ && index == label_idx_;
                if (!number(ri.ptr, c, v, is_label)) return false;
                if (is_label) {
                    class_set = true;
                    label = v;
                } else {
                    emit(has_label_ && index > label_idx_ ? index - 1 : index, v);
                }
                if (c >= e) break;
                p = skip_spaces(c + 1, e);
                if (p >= e) return false;  // a trailing comma
            }
            if (has_label_ && !class_set) return false;
        } else {  // dense: one value per attribute
            if (e[-1] == '}') return false;
            std::size_t attribute = 0, feature = 0;
            while (true) {
                const char *c = p;
                while (c < e && *c != ',') ++c;
                if (attribute >= num_attributes) return false;
                double v = 0.0;
                const bool is_label = has_label_ && attribute == label_idx_;
                if (!number(p, c, v, is_label)) return false;
                if (is_label) {
                    label = v;
                } else {
                    emit(feature++, v);
                }
                ++attribute;
                if (c >= e) break;
                p = c + 1;
            }
            if (attribute != num_attributes) return false;
        }
        return !has_label_ || allowed(label);
    }

    unsigned num_threads() const {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const std::size_t by_size = std::max<std::size_t>(1, num_points() / 2048);
        return static_cast<unsigned>(std::min({ static_cast<std::size_t>(hw), std::size_t(32), by_size }));
    }

    template <typename F>
    void run_parallel(unsigned nt, F &&body) const {
        const std::size_t n = num_points();
        if (nt <= 1) {
            body(std::size_t(0), n);
            return;
        }
        std::vector<std::thread> pool;
        pool.reserve(nt);
        for (unsigned t = 0; t < nt; ++t) {
            const std::size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            pool.emplace_back([&body, lo, hi] { body(lo, hi); });
        }
        for (std::thread &th : pool) th.join();
    }

    std::string text_;
    std::vector<Line> lines_;
    std::vector<double> allowed_;  // the class labels of the header, sorted
    std::size_t first_row_ = 0, num_features_ = 0, label_idx_ = 0;
    bool has_label_ = false, int_labels_ = false;
};

}  // namespace lssvm

#endif  // PLSSVM_AMD_ARFF_READER_HPP_
