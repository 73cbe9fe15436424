/*
 * libsvm_reader.hpp -- multi-threaded reader for well-formed LIBSVM data files (host code only).
 *
 * The format rules are the reference's (citations relative to /root/reference):
 *   - lines end at '\r' or '\n', are left-trimmed, and are dropped when empty or starting with '#'
 *     (src/plssvm/detail/io/file_reader.cpp:179-205);
 *   - a line starts with a label iff its first ':' comes after its first blank (include/plssvm/detail/io/libsvm_parsing.hpp:47-95);
 *     either every line carries a label or none does;
 *   - features are "index:value" with one-based, strictly increasing indices; missing features are zeros; the number of
 *     features is the largest index of the file (libsvm_parsing.hpp:118-229).
 * This reader is the FAST PATH for files that follow those rules to the letter.  Anything else -- a token that does not
 * convert, an index out of order, an in-line comment, mixed labelling -- makes it report failure without a diagnosis; the
 * caller (plssvm_amd/io_libsvm.py) then re-parses with the line-by-line Python implementation, which raises the reference's
 * exact error messages.  So the accepted language here may be narrower than the format, never wider.
 */
#ifndef PLSSVM_AMD_LIBSVM_READER_HPP_
#define PLSSVM_AMD_LIBSVM_READER_HPP_

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "text_file.hpp"

namespace lssvm {

class LibsvmFile {
  public:
    /* maps the file and indexes its data lines; returns false if the file cannot be read */
    bool open(const char *path, std::uint64_t skipped_lines) {
        if (!text_.open(path)) return false;
        index(0, skipped_lines);
        return true;
    }
    /* the data lines of an opened text from offset `from` on (model files: the lines after "SV", model_io.hpp) */
    void index(std::size_t from, std::uint64_t skipped_lines) { lines_ = text_.index_lines(from, '#', skipped_lines); }
    TextFile &text() { return text_; }

    /* pass 1: validates the structure, finds the number of features and whether the lines are labelled */
    bool scan() {
        if (lines_.empty()) return false;
        const unsigned nt = num_threads();
        std::vector<std::size_t> max_index(nt, 0);
        std::vector<int> labelled(nt, -1);  // -1 no line seen, 0 unlabelled, 1 labelled, 2 mixed
        std::atomic<bool> ok{ true };
        run_parallel(nt, [&](unsigned t, std::size_t lo, std::size_t hi) {
            std::size_t mx = 0;
            int lab = -1;
            for (std::size_t i = lo; i < hi && ok.load(std::memory_order_relaxed); ++i) {
                bool has_label = false;
                std::size_t last = 0;
                if (!walk_line(i, has_label, last, [](std::size_t, double) {}, nullptr)) {
                    ok.store(false, std::memory_order_relaxed);
                    return;
                }
                mx = std::max(mx, last);
                const int l = has_label ? 1 : 0;
                lab = (lab == -1 || lab == l) ? l : 2;
            }
            max_index[t] = mx;
            labelled[t] = lab;
        });
        if (!ok.load()) return false;
        num_features_ = 0;
        int lab = -1;
        for (unsigned t = 0; t < nt; ++t) {
            num_features_ = std::max(num_features_, max_index[t]);
            if (labelled[t] == -1) continue;
            lab = (lab == -1 || lab == labelled[t]) ? labelled[t] : 2;
        }
        if (num_features_ == 0 || lab == 2 || lab == -1) return false;
        has_label_ = lab == 1;
        return true;
    }

    /* pass 2: dense row-major matrix (leading dimension ldx >= num_features, zeroed here) and the labels (always double) */
    template <typename T>
    bool fill(T *X, std::size_t ldx, double *labels) const {
        if (ldx < num_features_) return false;
        std::atomic<bool> ok{ true };
        run_parallel(num_threads(), [&](unsigned, std::size_t lo, std::size_t hi) {
            for (std::size_t i = lo; i < hi; ++i) {
                T *row = X + i * ldx;
                std::fill(row, row + ldx, T(0));
                bool has_label = false;
                std::size_t last = 0;
                double label = 0.0;
                if (!walk_line(i, has_label, last, [&](std::size_t index, double v) { row[index - 1] = static_cast<T>(v); }, &label)) {
                    ok.store(false, std::memory_order_relaxed);
                    return;
                }
                if (labels != nullptr && has_label) labels[i] = label;
            }
        });
        return ok.load();
    }

    std::size_t num_points() const { return lines_.size(); }
    std::size_t num_features() const { return num_features_; }
    bool has_label() const { return has_label_; }

  private:
    using Line = TextFile::Line;

    static bool is_blank(char c) { return c == ' '; }  // tokens are separated by spaces; a tab makes the line "not well formed" here

    /* walks one line: label (if any), then index:value tokens; `emit(index, value)` per feature.  false = not well formed. */
    template <typename Emit>
    bool walk_line(std::size_t i, bool &has_label, std::size_t &last_index, Emit &&emit, double *label_out) const {
        const char *p = text_.data() + lines_[i].begin;
        const char *e = text_.data() + lines_[i].end;
        while (e > p && is_blank(e[-1])) --e;  // right trim
        // label: the first token, if it holds no ':'
        const char *tok_end = p;
        while (tok_end < e && !is_blank(*tok_end)) ++tok_end;
        bool colon = false;
        for (const char *c = p; c < tok_end; ++c) {
            if (*c == '#') return false;  // in-line comments are left to the reference-exact parser
            colon = colon || *c == ':';
        }
        has_label = !colon;
        if (has_label) {
            double v = 0.0;
            const auto r = std::from_chars(p, tok_end, v);
            if (r.ec != std::errc() || r.ptr != tok_end) return false;
            if (label_out != nullptr) *label_out = v;
            p = tok_end;
        }
        last_index = 0;
        while (true) {
            while (p < e && is_blank(*p)) ++p;
            if (p >= e) break;
            std::size_t index = 0;
            const auto ri = std::from_chars(p, e, index);
            if (ri.ec != std::errc() || ri.ptr >= e || *ri.ptr != ':') return false;
            if (index == 0 || index <= last_index) return false;
            p = ri.ptr + 1;
            double v = 0.0;
            const auto rv = std::from_chars(p, e, v);
            if (rv.ec != std::errc() || rv.ptr == p) return false;
            if (rv.ptr < e && !is_blank(*rv.ptr)) return false;  // trailing garbage, '#', ...
            p = rv.ptr;
            last_index = index;
            emit(index, v);
        }
        return true;
    }

    unsigned num_threads() const { return io_threads(lines_.size(), 2048); }

    template <typename F>
    void run_parallel(unsigned nt, F &&body) const {
        io_parallel(nt, lines_.size(), body);
    }

    TextFile text_;
    std::vector<Line> lines_;
    std::size_t num_features_ = 0;
    bool has_label_ = false;
};

}  // namespace lssvm

#endif  // PLSSVM_AMD_LIBSVM_READER_HPP_
