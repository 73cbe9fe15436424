/*
 * test_reader_sanitized.cpp -- host-side sanitizer pass over the native LIBSVM reader (plssvm_amd/csrc/libsvm_reader.hpp), CPU build only
 * (g++ -fsanitize=address,undefined).  The reader is the fast path for WELL-FORMED files: anything else must make it report
 * failure -- never read or write out of bounds, never overflow an index.  Inputs: the shapes of the reference's own invalid fixtures
 * (the files under /root/reference/tests/data/libsvm/invalid, restated here as data), every truncation of a small valid file, huge and overflowing indices, empty
 * files, files without a final newline, CR / CRLF line ends, comments, tabs, NUL bytes, very long lines, label-only lines.  The format rules it
 * must agree with: /root/reference/include/plssvm/detail/io/libsvm_parsing.hpp:47-95, :118-229.
 * Exit code 0 = every case behaved (valid files parse to the expected shape and values, invalid ones are refused), and the sanitizers stayed silent.
 */
#include "../../plssvm_amd/csrc/libsvm_reader.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static int failures = 0;
static std::string tmp_path;

static void write_file(const std::string &content) {
    std::FILE *f = std::fopen(tmp_path.c_str(), "wb");
    if (f == nullptr) {
        std::perror("fopen");
        std::exit(2);
    }
    if (!content.empty()) std::fwrite(content.data(), 1, content.size(), f);
    std::fclose(f);
}

struct Parsed {
    bool ok = false;
    std::size_t points = 0, features = 0;
    bool labelled = false;
    std::vector<double> X, y;
};

/* open + scan + fill, the sequence of lssvm_mi355_libsvm_open / _fill_f64 (capi.hip); fill only where the dense matrix is of test size */
static Parsed parse(const std::string &content, std::uint64_t skipped = 0) {
    write_file(content);
    Parsed r;
    lssvm::LibsvmFile file;
    if (!file.open(tmp_path.c_str(), skipped) || !file.scan()) return r;
    r.points = file.num_points();
    r.features = file.num_features();
    r.labelled = file.has_label();
    if (r.points * r.features > (std::size_t(1) << 22)) {  // a caller would have to allocate this: refused by the binding, not filled here
        r.ok = true;
        return r;
    }
    r.X.assign(r.points * r.features, -7.0);
    r.y.assign(r.points, -7.0);
    r.ok = file.fill(r.X.data(), r.features, r.y.data());
    // a leading dimension that is too small must be refused, not overrun
    if (r.features > 1) {
        std::vector<double> small(r.points * (r.features - 1));
        if (file.fill(small.data(), r.features - 1, nullptr)) {
            std::printf("FAIL: fill accepted a leading dimension below the number of features\n");
            ++failures;
        }
    }
    return r;
}

static void expect_refused(const char *name, const std::string &content) {
    const Parsed r = parse(content);
    if (r.ok) {
        std::printf("FAIL: %s was accepted (%zu x %zu)\n", name, r.points, r.features);
        ++failures;
    }
}

static void expect_shape(const char *name, const std::string &content, std::size_t points, std::size_t features, bool labelled, std::uint64_t skipped = 0) {
    const Parsed r = parse(content, skipped);
    if (!r.ok || r.points != points || r.features != features || r.labelled != labelled) {
        std::printf("FAIL: %s: ok %d, %zu x %zu labelled %d (wanted %zu x %zu labelled %d)\n", name, r.ok ? 1 : 0, r.points, r.features, r.labelled ? 1 : 0, points, features, labelled ? 1 : 0);
        ++failures;
    }
}

int main() {
    const char *dir = std::getenv("TMPDIR");
    tmp_path = std::string(dir != nullptr ? dir : "/tmp") + "/plssvm_amd_reader_sanitized_" + std::to_string(static_cast<long>(std::rand())) + ".libsvm";

    const std::string valid = "1 1:-1.11 2:-2.90 4:0.5\n-1 2:-0.52 3:-0.33\n1 1:11.21 4:3.14e1\n";
    {
        const Parsed r = parse(valid);
        const double want[12] = { -1.11, -2.90, 0.0, 0.5, 0.0, -0.52, -0.33, 0.0, 11.21, 0.0, 0.0, 31.4 };
        bool same = r.ok && r.points == 3 && r.features == 4 && r.labelled && r.y[0] == 1.0 && r.y[1] == -1.0 && r.y[2] == 1.0;
        for (int i = 0; same && i < 12; ++i) same = std::fabs(r.X[i] - want[i]) < 1e-15;
        if (!same) {
            std::printf("FAIL: the valid file did not parse to the expected matrix\n");
            ++failures;
        }
    }
    // the shapes of the reference's invalid fixtures (tests/data/libsvm/invalid/)
    expect_refused("feature_with_alpha_char_at_the_beginning", "1 1:a-1.11 2:-2.90\n0 1:-0.52 2:-0.33\n");
    expect_refused("inconsistent_label_specification", "1 1:-1.11 2:-2.90\n 1:-0.52 2:-0.33\n0 1:11.21 2:3.14\n");
    expect_refused("index_with_alpha_char_at_the_beginning", "1 1:-1.11 !2:-2.90\n0 1:-0.52 2:-0.33\n");
    expect_refused("invalid_colon_at_the_beginning", ":1 1:-1.11 2:-2.90\n0 1:-0.52 2:-0.33\n");
    expect_refused("invalid_colon_in_the_middle", "1 1:-1.11 :2:-2.90\n0 1:-0.52 2:-0.33\n");
    expect_refused("missing_feature_value", "1 1:-1.11 2: 3:42.0\n0 1:-0.52 2:-0.33\n");
    expect_refused("missing_index_value", "1 1:-1.11 :-2.90 3:42.0\n0 1:-0.52 2:-0.33\n");
    expect_refused("non_increasing_indices", "1 1:-1.11 2:-2.90 3:187\n0 1:-0.52 3:-0.33 3:127.12\n");
    expect_refused("non_strictly_increasing_indices", "1 1:-1.11 2:-2.90 3:187\n0 1:-0.52 3:-0.33 2:127.12\n");
    expect_refused("zero_based_features", "1 0:-1.11 1:-2.90\n1 0:-0.52 1:-0.33\n");
    // empty and degenerate files
    expect_refused("empty file", "");
    expect_refused("only blank lines", "\n\n  \n\r\n");
    expect_refused("only comments", "# a\n#b\n");
    expect_refused("labels without features", "1\n-1\n");
    expect_refused("NUL bytes", std::string("1 1:1.0\n\0\0\0 2:3\n", 16));
    expect_refused("tab separated", "1\t1:1.0\t2:2.0\n");
    expect_refused("in-line comment", "1 1:1.0 # note\n");
    expect_refused("value with trailing garbage", "1 1:1.0x 2:2.0\n");
    expect_refused("negative index", "1 -1:1.0\n");
    expect_refused("index overflow (2^64)", "1 18446744073709551616:1.0\n");
    expect_refused("index overflow (20 digits)", "1 99999999999999999999:1.0\n");
    // huge but representable indices: accepted by the scan (the binding refuses the allocation), never filled here
    expect_shape("index 2^32", "1 4294967296:1.0\n", 1, std::size_t(1) << 32, true);
    expect_shape("index 2^63", "1 9223372036854775808:2.5\n", 1, std::size_t(1) << 63, true);
    // line ends, final line without newline, unlabelled files, skipped header lines
    expect_shape("CRLF", "1 1:1.0 2:2.0\r\n-1 1:3.0\r\n", 2, 2, true);
    expect_shape("CR only", "1 1:1.0\r-1 2:3.0\r", 2, 2, true);
    expect_shape("no final newline", "1 1:1.0\n-1 3:2.0", 2, 3, true);
    expect_shape("unlabelled", "1:1.0 2:2.0\n2:3.0\n", 2, 2, false);
    expect_shape("comment lines and leading blanks", "# header\n   1 1:1.0\n\n#x\n-1 2:1\n", 2, 2, true);
    expect_shape("skipped lines", "1 1:9 2:9 3:9\n1 1:1.0\n-1 2:1\n", 2, 2, true, 1);
    expect_shape("trailing blanks", "1 1:1.0   \n-1 2:1 \n", 2, 2, true);
    {
        std::string longline = "1";
        for (int i = 1; i <= 20000; ++i) longline += " " + std::to_string(i) + ":0.5";
        expect_shape("very long line", longline + "\n", 1, 20000, true);
    }
    {
        std::string many;  // enough lines for the multi-threaded passes (2048 lines per thread)
        for (int i = 0; i < 10000; ++i) many += (i % 2 ? "1 " : "-1 ") + std::to_string(1 + i % 7) + ":" + std::to_string(i) + ".25 9:1\n";
        expect_shape("10 000 lines (threads)", many, 10000, 9, true);
        many += "1 3:1 2:1\n";  // one bad line at the very end of the last thread's range
        expect_refused("10 001 lines, the last one out of order", many);
    }
    // every truncation of the valid file: accepted or refused, never out of bounds
    for (std::size_t cut = 0; cut <= valid.size(); ++cut) (void) parse(valid.substr(0, cut));
    // ... and every single-byte corruption of it with a few hostile bytes
    for (std::size_t pos = 0; pos < valid.size(); ++pos) {
        for (const char c : { ':', ' ', '\n', '#', '-', 'e', '\0', '9' }) {
            std::string s = valid;
            s[pos] = c;
            (void) parse(s);
        }
    }
    std::remove(tmp_path.c_str());
    std::printf("%s: %d failure(s)\n", failures == 0 ? "OK" : "FAILED", failures);
    return failures == 0 ? 0 : 1;
}
