/*
 * test_arff_reader_sanitized.cpp -- host-side sanitizer pass over the native ARFF reader (plssvm_amd/csrc/arff_reader.hpp), CPU build only
 * (g++ -fsanitize=address,undefined), the sibling of test_reader_sanitized.cpp.  The reader is the fast path for WELL-FORMED files: anything else must make
 * it report failure -- never read or write out of bounds.  Inputs: the shapes of the reference's own invalid ARFF fixtures (the files under
 * /root/reference/tests/data/arff/invalid, restated here as data), every truncation and every single-byte corruption of a small valid file, empty files, files
 * without a final newline, CR / CRLF line ends, NUL bytes, very long rows, index overflow in sparse rows, 10 000 rows through the threaded passes.  The format rules it
 * must agree with: /root/reference/include/plssvm/detail/io/arff_parsing.hpp:57-170, :196-372.
 * Exit code 0 = every case behaved (valid files parse to the expected shape and values, invalid ones are refused), and the sanitizers stayed silent.
 */
#include "../../plssvm_amd/csrc/arff_reader.hpp"

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

/* open + scan + fill, the sequence of lssvm_mi355_arff_open / _fill_f64 (capi.hip) */
static Parsed parse(const std::string &content, bool int_labels = false) {
    write_file(content);
    Parsed r;
    lssvm::ArffFile file;
    if (!file.open(tmp_path.c_str()) || !file.scan(int_labels)) return r;
    r.points = file.num_points();
    r.features = file.num_features();
    r.labelled = file.has_label();
    r.X.assign(r.points * r.features, -7.0);
    r.y.assign(r.points, -7.0);
    r.ok = file.fill(r.X.data(), r.features, r.y.data());
    if (r.features > 1) {  // a leading dimension that is too small must be refused, not overrun
        std::vector<double> small(r.points * (r.features - 1));
        if (file.fill(small.data(), r.features - 1, nullptr)) {
            std::printf("FAIL: fill accepted a leading dimension below the number of features\n");
            ++failures;
        }
    }
    return r;
}

static void expect_refused(const char *name, const std::string &content, bool int_labels = false) {
    const Parsed r = parse(content, int_labels);
    if (r.ok) {
        std::printf("FAIL: %s was accepted (%zu x %zu)\n", name, r.points, r.features);
        ++failures;
    }
}

static void expect_shape(const char *name, const std::string &content, std::size_t points, std::size_t features, bool labelled) {
    const Parsed r = parse(content);
    if (!r.ok || r.points != points || r.features != features || r.labelled != labelled) {
        std::printf("FAIL: %s: ok %d, %zu x %zu labelled %d (wanted %zu x %zu labelled %d)\n", name, r.ok ? 1 : 0, r.points, r.features, r.labelled ? 1 : 0, points, features, labelled ? 1 : 0);
        ++failures;
    }
}

int main() {
    const char *dir = std::getenv("TMPDIR");
    tmp_path = std::string(dir != nullptr ? dir : "/tmp") + "/plssvm_amd_arff_sanitized_" + std::to_string(static_cast<long>(std::rand())) + ".arff";

    const std::string header = "% a comment\n@RELATION test\n@ATTRIBUTE first NUMERIC\n@ATTRIBUTE second numeric\n@ATTRIBUTE class {-1,1}\n@ATTRIBUTE third NUMERIC\n@DATA\n";
    const std::string valid = header + "-1.11,-2.90,1,0.5\n{1 -0.52,2 -1,3 -0.33}\n11.21,0,1,3.14e1\n{2 1}\n";
    {
        const Parsed r = parse(valid);
        const double want[12] = { -1.11, -2.90, 0.5, 0.0, -0.52, -0.33, 11.21, 0.0, 31.4, 0.0, 0.0, 0.0 };
        bool same = r.ok && r.points == 4 && r.features == 3 && r.labelled && r.y[0] == 1.0 && r.y[1] == -1.0 && r.y[2] == 1.0 && r.y[3] == 1.0;
        for (int i = 0; same && i < 12; ++i) same = std::fabs(r.X[i] - want[i]) < 1e-15;
        if (!same) {
            std::printf("FAIL: the valid file did not parse to the expected matrix\n");
            ++failures;
        }
        if (!parse(valid, true).ok) {
            std::printf("FAIL: integer labels were refused with int_labels\n");
            ++failures;
        }
    }
    // the shapes of the reference's invalid fixtures (tests/data/arff/invalid/)
    expect_refused("@_inside_data_section", header + "1,2,1,3\n@ATTRIBUTE invalid numeric\n");
    expect_refused("class_same_label_multiple_times", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class {1,1}\n@DATA\n1,1\n");
    expect_refused("class_unquoted_nominal_attribute", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class    0,1\n@DATA\n1,1\n");
    expect_refused("class_with_only_one_label", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class {1}\n@DATA\n1,1\n");
    expect_refused("class_with_wrong_label (NUMERIC)", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class NUMERIC\n@DATA\n1,1\n");
    expect_refused("class_without_label", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class\n@DATA\n1,1\n");
    expect_refused("dense_missing_value", header + "1,2,1\n");
    expect_refused("dense_too_many_values", header + "1,2,1,3,4,5\n");
    expect_refused("multiple_classes", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class {0,1}\n@ATTRIBUTE class {0,1}\n@DATA\n1,1,1\n");
    expect_refused("no_data_attribute", "@RELATION t\n@ATTRIBUTE a NUMERIC\n1\n2\n");
    expect_refused("no_features", "@RELATION t\n@ATTRIBUTE class {0,1}\n@DATA\n1\n");
    expect_refused("nominal_attribute_with_wrong_name", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE foo    {0,1}\n@DATA\n1,1\n");
    expect_refused("numeric_unquoted", "@RELATION t\n@ATTRIBUTE second entry   numeric\n@DATA\n1\n");
    expect_refused("numeric_without_name", "@RELATION t\n@ATTRIBUTE   numeric\n@DATA\n1\n");
    expect_refused("relation_not_at_beginning", "@ATTRIBUTE a NUMERIC\n@RELATION t\n@DATA\n1\n");
    expect_refused("relation_unquoted", "@RELATION  name with whitespaces\n@ATTRIBUTE a NUMERIC\n@DATA\n1\n");
    expect_refused("relation_without_name", "@RELATION\n@ATTRIBUTE a NUMERIC\n@DATA\n1\n");
    expect_refused("sparse_invalid_feature_index", header + "{5 1.0,2 1}\n");
    expect_refused("sparse_missing_closing_brace", header + "{2 1,0 0.51\n");
    expect_refused("sparse_missing_label", header + "{0 1.88,1 2}\n");
    expect_refused("sparse_missing_opening_brace", header + "1 0.6,2 1}\n");
    expect_refused("string label", header + "1,2,foo,3\n");
    expect_refused("usage_of_undefined_label", header + "1,2,2,3\n");
    expect_refused("wrong_line", "@RELATION t\n@THIS IS NOT A CORRECT LINE\n@ATTRIBUTE a NUMERIC\n@DATA\n1\n");
    // empty and degenerate files, hostile bytes
    expect_refused("empty file", "");
    expect_refused("only blank lines", "\n\n  \n\r\n");
    expect_refused("only comments", "% a\n%b\n");
    expect_refused("header without rows", header);
    expect_refused("NUL bytes", header + std::string("1,2,1,3\n\0\0\0,4\n", 15));
    expect_refused("sparse index overflow (2^64)", header + "{18446744073709551616 1.0,2 1}\n");
    expect_refused("sparse index overflow (20 digits)", header + "{99999999999999999999 1.0,2 1}\n");
    expect_refused("a lone opening brace", header + "{\n");
    expect_refused("a lone closing brace", header + "}\n");
    expect_refused("value with trailing garbage", header + "1,2x,1,3\n");
    expect_refused("float label with int_labels", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE class {0,1.5}\n@DATA\n1,1.5\n", true);
    // line ends, final line without newline, files without a class attribute, keywords in any case
    expect_shape("CRLF", "@RELATION t\r\n@ATTRIBUTE a NUMERIC\r\n@ATTRIBUTE b NUMERIC\r\n@DATA\r\n1,2\r\n{1 3}\r\n", 2, 2, false);
    expect_shape("CR only", "@RELATION t\r@ATTRIBUTE a NUMERIC\r@DATA\r1\r2\r", 2, 1, false);
    expect_shape("no final newline", header + "1,2,-1,3", 1, 3, true);
    expect_shape("lower case keywords, blanks around values", "@relation t\n@attribute a numeric\n@attribute class {0, 1}\n@data\n 1.5 , 0\n{0 2,  1 1}\n", 2, 1, true);
    expect_shape("empty sparse row without a class attribute", "@RELATION t\n@ATTRIBUTE a NUMERIC\n@DATA\n{}\n1\n", 2, 1, false);
    {
        std::string head = "@RELATION wide\n", row;
        for (int i = 0; i < 20000; ++i) {
            head += "@ATTRIBUTE f" + std::to_string(i) + " NUMERIC\n";
            row += (i ? "," : "") + std::string("0.5");
        }
        expect_shape("very long row", head + "@DATA\n" + row + "\n", 1, 20000, false);
    }
    {
        std::string many = header;  // enough rows for the multi-threaded passes (2048 rows per thread)
        for (int i = 0; i < 10000; ++i) many += (i % 3 == 0) ? "{0 " + std::to_string(i) + ".25,2 -1}\n" : std::to_string(i) + ",0.5,1,-2e-3\n";
        expect_shape("10 000 rows (threads)", many, 10000, 3, true);
        many += "1,2,7,3\n";  // one bad row at the very end of the last thread's range
        expect_refused("10 001 rows, the last one with an unknown label", many);
    }
    // every truncation of the valid file: accepted or refused, never out of bounds
    for (std::size_t cut = 0; cut <= valid.size(); ++cut) (void) parse(valid.substr(0, cut));
    // ... and every single-byte corruption of it with a few hostile bytes
    for (std::size_t pos = 0; pos < valid.size(); ++pos) {
        for (const char c : { ',', ' ', '\n', '%', '-', 'e', '\0', '9', '{', '}', '@' }) {
            std::string s = valid;
            s[pos] = c;
            (void) parse(s);
        }
    }
    std::remove(tmp_path.c_str());
    std::printf("%s: %d failure(s)\n", failures == 0 ? "OK" : "FAILED", failures);
    return failures == 0 ? 0 : 1;
}
