/*
 * test_model_io_sanitized.cpp -- host-side sanitizer pass over the native model reader and the model / data writers (plssvm_amd/csrc/model_io.hpp,
 * text_file.hpp), CPU build only (g++ -fsanitize=address,undefined,thread-free: ASan + UBSan).  The reader is the fast path for WELL-FORMED model files:
 * anything else must make it report failure -- never read or write out of bounds; the writers format into per-thread buffers sized from a bound per entry
 * (RowWriter::kEntryBound): the longest things a row can hold (20-digit indices, three-digit exponents, denormals, non-finite values, long labels) must fit.
 * The format rules: /root/reference/include/plssvm/detail/io/libsvm_model_parsing.hpp:64-262 (reader), :296-499 (writer),
 * /root/reference/include/plssvm/detail/io/libsvm_parsing.hpp:244-296 (data writer).
 * Exit code 0 = every case behaved and the sanitizers stayed silent.
 */
#include "../../plssvm_amd/csrc/model_io.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

static int failures = 0;
static std::string tmp_path;

static void fail(const char *what) {
    std::printf("FAIL: %s\n", what);
    ++failures;
}

static std::string slurp() {
    std::string out;
    std::FILE *f = std::fopen(tmp_path.c_str(), "rb");
    if (f == nullptr) return out;
    char buf[4096];
    std::size_t got = 0;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, got);
    std::fclose(f);
    return out;
}

struct Loaded {
    bool ok = false;
    std::size_t sv = 0, features = 0;
    std::vector<double> X, alpha;
    lssvm::ModelHeader header;
};

static Loaded load_text(const std::string &text) {
    Loaded r;
    lssvm::ModelFile file;
    if (!file.open_text(text)) return r;
    r.sv = file.num_support_vectors();
    r.features = file.num_features();
    r.header = file.header();
    if (r.sv * r.features > (std::size_t(1) << 22)) return r;
    r.X.assign(r.sv * r.features, -7.0);
    r.alpha.assign(r.sv, -7.0);
    r.ok = file.fill<double>(r.X.data(), r.features, r.alpha.data());
    if (r.features > 1) {
        std::vector<double> small(r.sv * (r.features - 1));
        if (file.fill<double>(small.data(), r.features - 1, r.alpha.data())) fail("fill accepted a leading dimension below the number of features");
    }
    return r;
}

static void expect_refused(const char *name, const std::string &text) {
    if (load_text(text).ok) {
        std::printf("FAIL: %s was accepted\n", name);
        ++failures;
    }
}

int main() {
    const char *dir = std::getenv("TMPDIR");
    tmp_path = std::string(dir != nullptr ? dir : "/tmp") + "/plssvm_amd_model_io_sanitized_" + std::to_string(static_cast<long>(::getpid())) + ".model";

    const std::string header = "svm_type c_svc\nkernel_type polynomial\ndegree 3\ngamma 0.5\ncoef0 1.5\nnr_class 2\ntotal_sv 3\nrho 0.25\nlabel 1 -1\nnr_sv 2 1\nSV\n";
    const std::string body = "1.0e+00 1:1.0e+00 3:2.0e+00 \n-5.0e-01 2:1.0e+00 \n2.5e-01 1:3.0e+00 2:-1.0e+00 \n";
    {
        const Loaded r = load_text("# c\n" + header + body);
        const double want[9] = { 1, 0, 2, 0, 1, 0, 3, -1, 0 };
        bool same = r.ok && r.sv == 3 && r.features == 3 && r.header.kernel_type == 1 && r.header.degree == 3 && r.header.gamma == 0.5 && r.header.coef0 == 1.5
                    && r.header.rho == 0.25 && r.header.labels.size() == 2 && r.header.labels[1] == "-1" && r.header.nr_sv[0] == 2 && r.alpha[1] == -0.5;
        for (int i = 0; same && i < 9; ++i) same = r.X[i] == want[i];
        if (!same) fail("the valid model did not load as expected");
    }
    expect_refused("empty text", "");
    expect_refused("header only", header);
    expect_refused("no SV line", header.substr(0, header.size() - 3) + body);
    expect_refused("NUL in the header", std::string("svm_type c_svc\nkernel_type\0 rbf\n", 32) + body);
    expect_refused("high bytes in the header", "svm_type c_svc\nkernel_type rbf\nlabel \xc3\xa4 b\n" + body);
    expect_refused("a key without a value", "svm_type\n" + header + body);
    expect_refused("total_sv overflows", "svm_type c_svc\nkernel_type linear\nnr_class 2\ntotal_sv 99999999999999999999\nrho 0\nlabel 1 -1\nnr_sv 1 1\nSV\n" + body);
    expect_refused("nr_sv wraps around", "svm_type c_svc\nkernel_type linear\nnr_class 2\ntotal_sv 3\nrho 0\nlabel 1 -1\nnr_sv 18446744073709551615 4\nSV\n" + body);
    expect_refused("fewer lines than total_sv", header + "1.0 1:1\n");
    expect_refused("a 100-line header", [&] {
        std::string t;
        for (int i = 0; i < 100; ++i) t += "# not a comment\nrho 1\n";
        return t + body;
    }());
    // every truncation and a set of single-byte corruptions of the valid model: accepted or refused, never out of bounds
    const std::string valid = header + body;
    for (std::size_t cut = 0; cut <= valid.size(); ++cut) (void) load_text(valid.substr(0, cut));
    for (std::size_t pos = 0; pos < valid.size(); ++pos) {
        for (const char c : { ':', ' ', '\n', '#', '-', 'e', '\0', '9', '\t', '\r' }) {
            std::string s = valid;
            s[pos] = c;
            (void) load_text(s);
        }
    }
    // the mapped-file path, with more lines than one thread indexes
    {
        std::string big = "svm_type c_svc\nkernel_type linear\nnr_class 2\ntotal_sv 30000\nrho 0\nlabel a b\nnr_sv 10000 20000\nSV\n";
        for (int i = 0; i < 30000; ++i) big += std::to_string(i) + ".5e-3 " + std::to_string(1 + i % 5) + ":" + std::to_string(i) + " 7:1 \n";
        std::FILE *f = std::fopen(tmp_path.c_str(), "wb");
        std::fwrite(big.data(), 1, big.size(), f);
        std::fclose(f);
        lssvm::ModelFile file;
        std::vector<float> X(30000 * 7), a(30000);
        if (!file.open(tmp_path.c_str()) || file.num_support_vectors() != 30000 || file.num_features() != 7 || !file.fill<float>(X.data(), 7, a.data())
            || a[29999] != 29999.5e-3f || X[29999 * 7 + 4] != 29999.0f)
            fail("the 30 000 line model did not load through the mapped file");
    }

    // ---- writers: the longest entries must fit the per-thread buffers ----
    for (const unsigned threads : { 1u, 3u, 8u }) {
        lssvm::io_thread_limit().store(threads);
        const std::size_t n = 5000, d = 9;
        std::vector<double> X(n * d), alpha(n);
        const double specials[] = { 0.0, -0.0, std::numeric_limits<double>::denorm_min(), -std::numeric_limits<double>::max(), std::numeric_limits<double>::infinity(),
                                    -std::numeric_limits<double>::infinity(), std::numeric_limits<double>::quiet_NaN(), 1e-300, -1.2345678901234567e+300 };
        for (std::size_t i = 0; i < n * d; ++i) X[i] = specials[(i * 7 + i / d) % 9];
        for (std::size_t i = 0; i < n; ++i) alpha[i] = specials[(i + 2) % 9];
        std::vector<std::uint64_t> order(n);
        for (std::size_t i = 0; i < n; ++i) order[i] = n - 1 - i;
        lssvm::RowWriter out;
        if (!out.open(tmp_path.c_str(), "H\n", 2) || !out.write_rows<double>(X.data(), d, d, order.data(), n, lssvm::AlphaPrefix<double>{ alpha.data() }) || !out.close())
            fail("model body writer failed");
        const std::string text = slurp();
        if (text.size() != out.bytes() || text.compare(0, 2, "H\n") != 0 || static_cast<std::size_t>(std::count(text.begin(), text.end(), '\n')) != n + 1)
            fail("model body writer: wrong size or line count");
        // labels as long text + a single huge row: the chunk size falls back to one row
        std::string labels;
        std::vector<std::uint64_t> offsets{ 0 };
        for (int i = 0; i < 3; ++i) {
            labels += std::string(1000 + i, static_cast<char>('a' + i));
            offsets.push_back(labels.size());
        }
        std::vector<float> wide(3 * 40000, -1.17549435e-38f);
        lssvm::RowWriter data;
        if (!data.open(tmp_path.c_str(), nullptr, 0) || !data.write_rows<float>(wide.data(), 40000, 40000, nullptr, 3, lssvm::TextPrefix{ labels.data(), offsets.data(), 1002 }) || !data.close())
            fail("data writer failed");
        if (slurp().size() != data.bytes()) fail("data writer: size mismatch");
        const std::int64_t ints[3] = { std::numeric_limits<std::int64_t>::min(), 0, std::numeric_limits<std::int64_t>::max() };
        lssvm::RowWriter idata;
        if (!idata.open(tmp_path.c_str(), nullptr, 0) || !idata.write_rows<float>(wide.data(), 2, 40000, nullptr, 3, lssvm::IntegerPrefix{ ints }) || !idata.close()) fail("integer-label writer failed");
        if (slurp().compare(0, 21, "-9223372036854775808 ") != 0) fail("integer-label writer: wrong text");
    }
    lssvm::io_thread_limit().store(0);
    {
        lssvm::RowWriter nowhere;
        if (nowhere.open("/nonexistent_dir_for_this_test/x.model", "h", 1) || nowhere.error() == 0) fail("opening an impossible path succeeded");
    }
    std::remove(tmp_path.c_str());
    std::printf("%s: %d failure(s)\n", failures == 0 ? "OK" : "FAILED", failures);
    return failures == 0 ? 0 : 1;
}
