/*
 * text_file.hpp -- a read-only text file in memory and the index of its lines, shared by the native LIBSVM data reader (libsvm_reader.hpp) and
 * the model reader (model_io.hpp).  Host code only.
 *
 * The line rules are the reference's file_reader (src/plssvm/detail/io/file_reader.cpp:179-205, citation relative to /root/reference): lines end at
 * '\r' or '\n', are left-trimmed, and are dropped when empty or starting with the comment character.  Like the reference (file_reader.cpp:84-120) the
 * file is memory mapped where the platform allows and read into a buffer otherwise; the index is built by all threads, each over its own piece of the
 * text cut at a line end.
 */
#ifndef PLSSVM_AMD_TEXT_FILE_HPP_
#define PLSSVM_AMD_TEXT_FILE_HPP_

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace lssvm {

/* upper bound of the worker threads of the readers and writers: 0 = the hardware's (test aid, lssvm_mi355_set_io_threads: a file must not depend on it) */
inline std::atomic<unsigned> &io_thread_limit() {
    static std::atomic<unsigned> limit{ 0 };
    return limit;
}

/* number of worker threads for `units` pieces of work of which one thread should get at least `grain` */
inline unsigned io_threads(std::size_t units, std::size_t grain) {
    const unsigned limit = io_thread_limit().load(std::memory_order_relaxed);
    const unsigned hw = limit != 0 ? limit : std::max(1u, std::thread::hardware_concurrency());
    const std::size_t by_size = std::max<std::size_t>(1, units / std::max<std::size_t>(1, grain));
    return static_cast<unsigned>(std::min({ static_cast<std::size_t>(hw), std::size_t(32), by_size }));
}

/* body(t, lo, hi) on nt threads over [0, n) cut into nt contiguous ranges */
template <typename F>
void io_parallel(unsigned nt, std::size_t n, F &&body) {
    if (nt <= 1) {
        body(0u, std::size_t(0), n);
        return;
    }
    // an exception of a worker (std::bad_alloc of its buffer) is carried to the caller instead of ending the process; a thread that cannot be started has its range run here
    std::vector<std::thread> pool;
    pool.reserve(nt);
    std::exception_ptr failure;
    std::mutex failure_mutex;
    auto guarded_body = [&](unsigned t, std::size_t lo, std::size_t hi) {
        try {
            body(t, lo, hi);
        } catch (...) {
            const std::lock_guard<std::mutex> lock(failure_mutex);
            if (!failure) failure = std::current_exception();
        }
    };
    for (unsigned t = 0; t < nt; ++t) {
        const std::size_t lo = n / nt * t + std::min<std::size_t>(t, n % nt), hi = n / nt * (t + 1) + std::min<std::size_t>(t + 1, n % nt);
        try {
            pool.emplace_back([&guarded_body, t, lo, hi] { guarded_body(t, lo, hi); });
        } catch (const std::system_error &) {
            guarded_body(t, lo, hi);
        }
    }
    for (std::thread &th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
}

class TextFile {
  public:
    struct Line {
        std::size_t begin, end;  // [begin, end) of the left-trimmed line, no line end
    };

    TextFile() = default;
    TextFile(const TextFile &) = delete;
    TextFile &operator=(const TextFile &) = delete;
    ~TextFile() { release(); }

    /* false if the file cannot be opened or read */
    bool open(const char *path) {
        release();
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return false;
        struct stat st {};
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            ::close(fd);
            return false;
        }
        size_ = static_cast<std::size_t>(st.st_size);
        if (size_ == 0) {
            ::close(fd);
            data_ = "";
            return true;
        }
        void *m = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            (void) ::madvise(m, size_, MADV_WILLNEED);
            data_ = static_cast<const char *>(m);
            mapped_ = true;
            ::close(fd);
            return true;
        }
        // no mapping (a file system without mmap): read it
        owned_.resize(size_);
        std::size_t got = 0;
        while (got < size_) {
            const ssize_t r = ::read(fd, &owned_[got], size_ - got);
            if (r <= 0) break;
            got += static_cast<std::size_t>(r);
        }
        ::close(fd);
        if (got != size_) return false;
        data_ = owned_.data();
        return true;
    }
    /* a text that is already in memory (tests) */
    void adopt(std::string text) {
        release();
        owned_ = std::move(text);
        data_ = owned_.data();
        size_ = owned_.size();
    }

    const char *data() const { return data_; }
    std::size_t size() const { return size_; }

    static bool is_left_blank(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }

    /* the first line at or after `from` that is neither empty nor a comment; false at the end of the text.  *next = where the search continues. */
    bool next_line(std::size_t from, char comment, Line &out, std::size_t &next) const {
        const char *b = data_, *e = data_ + size_;
        const char *p = b + std::min(from, size_);
        while (p < e) {
            const char *q = p;
            while (q < e && *q != '\n' && *q != '\r') ++q;
            const char *s = p;
            while (s < q && is_left_blank(*s)) ++s;
            p = q < e ? q + 1 : e;
            if (s < q && *s != comment) {
                out = { static_cast<std::size_t>(s - b), static_cast<std::size_t>(q - b) };
                next = static_cast<std::size_t>(p - b);
                return true;
            }
        }
        next = size_;
        return false;
    }

    /* index of the lines of [from, size) that are neither empty nor comments, the first `skipped` of them left out */
    std::vector<Line> index_lines(std::size_t from, char comment, std::uint64_t skipped) const {
        std::vector<Line> lines;
        from = std::min(from, size_);
        const std::size_t span = size_ - from;
        const unsigned nt = io_threads(span, std::size_t(4) << 20);
        // cut points: the first character after a line end at or after the even split
        std::vector<std::size_t> cut(nt + 1, size_);
        cut[0] = from;
        for (unsigned t = 1; t < nt; ++t) {
            std::size_t c = from + span / nt * t;
            while (c > from && c < size_ && data_[c - 1] != '\n' && data_[c - 1] != '\r') ++c;
            cut[t] = std::max(c, cut[t - 1]);
        }
        std::vector<std::vector<Line>> part(nt);
        io_parallel(nt, nt, [&](unsigned, std::size_t lo, std::size_t hi) {
            for (std::size_t t = lo; t < hi; ++t) {
                std::vector<Line> &mine = part[t];
                mine.reserve((cut[t + 1] - cut[t]) / 64 + 16);
                const char *b = data_, *e = data_ + cut[t + 1];
                for (const char *p = b + cut[t]; p < e;) {
                    const char *q = p;
                    while (q < e && *q != '\n' && *q != '\r') ++q;
                    const char *s = p;
                    while (s < q && is_left_blank(*s)) ++s;
                    if (s < q && *s != comment) mine.push_back({ static_cast<std::size_t>(s - b), static_cast<std::size_t>(q - b) });
                    p = q + 1;
                }
            }
        });
        std::size_t total = 0;
        for (const auto &v : part) total += v.size();
        if (skipped >= total) return lines;
        lines.reserve(total - skipped);
        std::uint64_t seen = 0;
        for (const auto &v : part) {
            for (const Line &l : v) {
                if (seen >= skipped) lines.push_back(l);
                ++seen;
            }
        }
        return lines;
    }

  private:
    void release() {
        if (mapped_) ::munmap(const_cast<char *>(data_), size_);
        mapped_ = false;
        data_ = nullptr;
        size_ = 0;
        owned_.clear();
        owned_.shrink_to_fit();
    }

    const char *data_ = nullptr;
    std::size_t size_ = 0;
    bool mapped_ = false;
    std::string owned_;
};

}  // namespace lssvm

#endif  // PLSSVM_AMD_TEXT_FILE_HPP_
