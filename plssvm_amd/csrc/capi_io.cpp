/*
 * capi_io.cpp -- the file-format entry points of libplssvm_amd.so (declared in include/plssvm_amd.h): the native readers for LIBSVM / ARFF data files and
 * LIBSVM model files, the writers for model files and LIBSVM data files.  Host code only: this translation unit includes no HIP header and needs no device
 * (the formats either side of the hot path).
 */
#include "lssvm_error.hpp"

#include "../../include/plssvm_amd_testing.h"

#include "arff_reader.hpp"
#include "libsvm_reader.hpp"
#include "model_io.hpp"

#include <cstring>
#include <memory>

using lssvm::guarded;

namespace {

template <typename T>
void model_write(const char *path, const char *header, const T *sv, uint64_t num_sv, uint64_t num_features, uint64_t ldx, const T *alpha, const uint64_t *order,
                 uint64_t count) {
    LSSVM_REQUIRE(path != nullptr, "path must not be NULL");
    LSSVM_REQUIRE(num_sv == 0 || (sv != nullptr && alpha != nullptr), "support_vectors / alpha must not be NULL");
    LSSVM_REQUIRE(ldx >= num_features, "ldx must be at least num_features");
    if (order == nullptr) {
        LSSVM_REQUIRE(count == num_sv, "without an order every support vector is written: count must equal num_support_vectors");
    } else {
        for (uint64_t k = 0; k < count; ++k) LSSVM_REQUIRE(order[k] < num_sv, "order holds a row index beyond num_support_vectors");
    }
    lssvm::RowWriter out;
    const bool ok = out.open(path, header, header != nullptr ? std::strlen(header) : 0)
                    && out.write_rows<T>(sv, num_features, ldx, order, count, lssvm::AlphaPrefix<T>{ alpha });
    if (!(out.close() && ok)) throw lssvm::Error(LSSVM_ERR_INTERNAL, std::string("can't write '") + path + "': " + std::strerror(out.error()));
}

template <typename T>
void data_write(const char *path, const char *header, const T *X, uint64_t num_points, uint64_t num_features, uint64_t ldx, const int64_t *int_labels,
                const char *label_text, const uint64_t *label_offsets) {
    LSSVM_REQUIRE(path != nullptr, "path must not be NULL");
    LSSVM_REQUIRE(num_points == 0 || X != nullptr, "X must not be NULL");
    LSSVM_REQUIRE(ldx >= num_features, "ldx must be at least num_features");
    LSSVM_REQUIRE(int_labels == nullptr || label_text == nullptr, "labels are given as integers OR as text, not both");
    LSSVM_REQUIRE((label_text == nullptr) == (label_offsets == nullptr), "label_text and label_offsets come together");
    lssvm::RowWriter out;
    bool ok = out.open(path, header, header != nullptr ? std::strlen(header) : 0);
    if (ok && int_labels != nullptr) {
        ok = out.write_rows<T>(X, num_features, ldx, nullptr, num_points, lssvm::IntegerPrefix{ int_labels });
    } else if (ok) {
        std::size_t longest = 0;
        if (label_text != nullptr) {
            for (uint64_t i = 0; i < num_points; ++i) {
                LSSVM_REQUIRE(label_offsets[i + 1] >= label_offsets[i], "label_offsets must not decrease");
                longest = std::max<std::size_t>(longest, label_offsets[i + 1] - label_offsets[i]);
            }
        }
        ok = out.write_rows<T>(X, num_features, ldx, nullptr, num_points, lssvm::TextPrefix{ label_text, label_offsets, longest });
    }
    if (!(out.close() && ok)) throw lssvm::Error(LSSVM_ERR_INTERNAL, std::string("can't write '") + path + "': " + std::strerror(out.error()));
}

}  // namespace

extern "C" {

/* test aid (include/plssvm_amd_testing.h): bound of the worker threads of the readers and writers, 0 = the hardware's */
int lssvm_mi355_set_io_threads(int threads) {
    return guarded([&] {
        LSSVM_REQUIRE(threads >= 0 && threads <= 1024, "threads must lie in [0, 1024]");
        lssvm::io_thread_limit().store(static_cast<unsigned>(threads));
    });
}

/* ---- LIBSVM data files: fast reader for well-formed files (libsvm_reader.hpp) ---- */
struct lssvm_mi355_libsvm_file {
    lssvm::LibsvmFile impl;
};

int lssvm_mi355_libsvm_open(const char *path, uint64_t skipped_lines, lssvm_mi355_libsvm_file **file_out, uint64_t *num_points, uint64_t *num_features,
                            int *has_label) {
    return guarded([&] {
        LSSVM_REQUIRE(path != nullptr && file_out != nullptr && num_points != nullptr && num_features != nullptr && has_label != nullptr,
                      "path / output pointers must not be NULL");
        *file_out = nullptr;
        auto f = std::make_unique<lssvm_mi355_libsvm_file>();
        if (!f->impl.open(path, skipped_lines)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, std::string("Couldn't find file: '") + path + "'!");
        if (!f->impl.scan()) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed LIBSVM data file for the fast reader");
        *num_points = f->impl.num_points();
        *num_features = f->impl.num_features();
        *has_label = f->impl.has_label() ? 1 : 0;
        *file_out = f.release();
    });
}
int lssvm_mi355_libsvm_fill_f32(lssvm_mi355_libsvm_file *file, float *X, uint64_t ldx, double *labels) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && X != nullptr, "file / X must not be NULL");
        if (!file->impl.fill<float>(X, ldx, labels)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed LIBSVM data file for the fast reader");
    });
}
int lssvm_mi355_libsvm_fill_f64(lssvm_mi355_libsvm_file *file, double *X, uint64_t ldx, double *labels) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && X != nullptr, "file / X must not be NULL");
        if (!file->impl.fill<double>(X, ldx, labels)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed LIBSVM data file for the fast reader");
    });
}
int lssvm_mi355_libsvm_close(lssvm_mi355_libsvm_file *file) {
    return guarded([&] { delete file; });
}

/* ---- LIBSVM data files: writer (model_io.hpp) ---- */
int lssvm_mi355_libsvm_write_f32(const char *path, const char *header, const float *X, uint64_t num_points, uint64_t num_features, uint64_t ldx,
                                 const int64_t *int_labels, const char *label_text, const uint64_t *label_offsets) {
    return guarded([&] { data_write<float>(path, header, X, num_points, num_features, ldx, int_labels, label_text, label_offsets); });
}
int lssvm_mi355_libsvm_write_f64(const char *path, const char *header, const double *X, uint64_t num_points, uint64_t num_features, uint64_t ldx,
                                 const int64_t *int_labels, const char *label_text, const uint64_t *label_offsets) {
    return guarded([&] { data_write<double>(path, header, X, num_points, num_features, ldx, int_labels, label_text, label_offsets); });
}

/* ---- LIBSVM model files: writer and fast reader (model_io.hpp) ---- */
int lssvm_mi355_model_write_f32(const char *path, const char *header, const float *support_vectors, uint64_t num_support_vectors, uint64_t num_features,
                                uint64_t ldx, const float *alpha, const uint64_t *order, uint64_t count) {
    return guarded([&] { model_write<float>(path, header, support_vectors, num_support_vectors, num_features, ldx, alpha, order, count); });
}
int lssvm_mi355_model_write_f64(const char *path, const char *header, const double *support_vectors, uint64_t num_support_vectors, uint64_t num_features,
                                uint64_t ldx, const double *alpha, const uint64_t *order, uint64_t count) {
    return guarded([&] { model_write<double>(path, header, support_vectors, num_support_vectors, num_features, ldx, alpha, order, count); });
}

struct lssvm_mi355_model_file {
    lssvm::ModelFile impl;
    std::string label_text;  // the labels of the header's "label" line, separated by single blanks
};

int lssvm_mi355_model_open(const char *path, lssvm_mi355_model_file **file_out, lssvm_model_info *info) {
    return guarded([&] {
        LSSVM_REQUIRE(path != nullptr && file_out != nullptr && info != nullptr, "path / output pointers must not be NULL");
        *file_out = nullptr;
        auto f = std::make_unique<lssvm_mi355_model_file>();
        if (!f->impl.open(path)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, std::string("'") + path + "' is not a well-formed LIBSVM model file for the fast reader");
        const lssvm::ModelHeader &h = f->impl.header();
        for (const std::string &l : h.labels) {
            if (!f->label_text.empty()) f->label_text += ' ';
            f->label_text += l;
        }
        *info = lssvm_model_info{};
        info->kernel_type = h.kernel_type;
        info->has_degree = h.has_degree ? 1 : 0;
        info->has_gamma = h.has_gamma ? 1 : 0;
        info->has_coef0 = h.has_coef0 ? 1 : 0;
        info->degree = h.degree;
        info->gamma = h.gamma;
        info->coef0 = h.coef0;
        info->rho = h.rho;
        info->nr_class = h.nr_class;
        info->total_sv = h.total_sv;
        info->num_features = f->impl.num_features();
        info->label_text_bytes = f->label_text.size() + 1;
        *file_out = f.release();
    });
}
int lssvm_mi355_model_labels(lssvm_mi355_model_file *file, char *label_text_out, uint64_t label_text_bytes, uint64_t *nr_sv_out) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr, "file must not be NULL");
        if (label_text_out != nullptr) {
            LSSVM_REQUIRE(label_text_bytes >= file->label_text.size() + 1, "label_text_out is shorter than lssvm_model_info.label_text_bytes");
            std::memcpy(label_text_out, file->label_text.c_str(), file->label_text.size() + 1);
        }
        if (nr_sv_out != nullptr) {
            const auto &nr_sv = file->impl.header().nr_sv;
            std::copy(nr_sv.begin(), nr_sv.end(), nr_sv_out);
        }
    });
}
int lssvm_mi355_model_fill_f32(lssvm_mi355_model_file *file, float *support_vectors, uint64_t ldx, float *alpha) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && support_vectors != nullptr && alpha != nullptr, "file / support_vectors / alpha must not be NULL");
        if (!file->impl.fill<float>(support_vectors, ldx, alpha)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed LIBSVM model file for the fast reader");
    });
}
int lssvm_mi355_model_fill_f64(lssvm_mi355_model_file *file, double *support_vectors, uint64_t ldx, double *alpha) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && support_vectors != nullptr && alpha != nullptr, "file / support_vectors / alpha must not be NULL");
        if (!file->impl.fill<double>(support_vectors, ldx, alpha)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed LIBSVM model file for the fast reader");
    });
}
int lssvm_mi355_model_close(lssvm_mi355_model_file *file) {
    return guarded([&] { delete file; });
}

/* ---- ARFF data files: fast reader for well-formed files (arff_reader.hpp) ---- */
struct lssvm_mi355_arff_file {
    lssvm::ArffFile impl;
};

int lssvm_mi355_arff_open(const char *path, int int_labels, lssvm_mi355_arff_file **file_out, uint64_t *num_points, uint64_t *num_features, int *has_label) {
    return guarded([&] {
        LSSVM_REQUIRE(path != nullptr && file_out != nullptr && num_points != nullptr && num_features != nullptr && has_label != nullptr,
                      "path / output pointers must not be NULL");
        *file_out = nullptr;
        auto f = std::make_unique<lssvm_mi355_arff_file>();
        if (!f->impl.open(path)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, std::string("Couldn't find file: '") + path + "'!");
        if (!f->impl.scan(int_labels != 0)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed ARFF data file for the fast reader");
        *num_points = f->impl.num_points();
        *num_features = f->impl.num_features();
        *has_label = f->impl.has_label() ? 1 : 0;
        *file_out = f.release();
    });
}
int lssvm_mi355_arff_fill_f32(lssvm_mi355_arff_file *file, float *X, uint64_t ldx, double *labels) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && X != nullptr, "file / X must not be NULL");
        if (!file->impl.fill<float>(X, ldx, labels)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed ARFF data file for the fast reader");
    });
}
int lssvm_mi355_arff_fill_f64(lssvm_mi355_arff_file *file, double *X, uint64_t ldx, double *labels) {
    return guarded([&] {
        LSSVM_REQUIRE(file != nullptr && X != nullptr, "file / X must not be NULL");
        if (!file->impl.fill<double>(X, ldx, labels)) throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "not a well-formed ARFF data file for the fast reader");
    });
}
int lssvm_mi355_arff_close(lssvm_mi355_arff_file *file) {
    return guarded([&] { delete file; });
}

}  // extern "C"
