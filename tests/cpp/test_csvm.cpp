/*
 * test_csvm.cpp -- C++ tests of the host-side adaptor (include/plssvm_amd/csvm.hpp) that read like the reference's own
 * backend tests (tests/backends/generic_csvm_tests.hpp, tests/csvm_factory.cpp), without GoogleTest (not in this image).
 *
 *   ./test_csvm            on a GPU box: runs everything            (exit code = number of failed checks)
 *   ./test_csvm --no-gpu   on a box without a GPU: only the checks that must work there (factory, exceptions, loud failure)
 */
#include "plssvm_amd/csvm.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <sstream>
#include <string>
#include <tuple>
#include <atomic>
#include <thread>
#include <vector>

namespace pa = plssvm_amd;

static int g_failed = 0;
static int g_checks = 0;
#define EXPECT_TRUE(cond)                                                          \
    do {                                                                           \
        ++g_checks;                                                                \
        if (!(cond)) {                                                             \
            ++g_failed;                                                            \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);           \
        }                                                                          \
    } while (0)
#define EXPECT_THROW_WHAT(stmt, extype, text)                                                        \
    do {                                                                                             \
        ++g_checks;                                                                                  \
        bool ok_ = false;                                                                            \
        try {                                                                                        \
            stmt;                                                                                    \
        } catch (const extype &e) {                                                                  \
            ok_ = std::string{ e.what() }.find(text) != std::string::npos;                           \
            if (!ok_) std::printf("  what(): %s\n", e.what());                                       \
        } catch (...) {                                                                              \
        }                                                                                            \
        if (!ok_) {                                                                                  \
            ++g_failed;                                                                              \
            std::printf("FAILED %s:%d: expected %s containing \"%s\"\n", __FILE__, __LINE__, #extype, text); \
        }                                                                                            \
    } while (0)

// EXPECT_FLOATING_POINT_NEAR of the reference (tests/custom_test_macros.hpp:114-137)
template <typename T>
static bool fp_near(T a, T b, T factor = T(128)) {
    if (a == b) return true;
    const T diff = std::abs(a - b);
    const T eps = std::numeric_limits<T>::epsilon();
    return diff < std::max(std::numeric_limits<T>::min(), factor * eps * (std::abs(a) + std::abs(b)));
}
template <typename T>
static bool fp_vec_near(const std::vector<T> &a, const std::vector<T> &b) {
    if (a.size() != b.size()) return false;
    for (std::size_t i = 0; i < a.size(); ++i)
        if (!fp_near(a[i], b[i])) return false;
    return true;
}

// re-exports the protected virtuals, exactly like tests/backends/HIP/mock_hip_csvm.hpp:22-44
class mock_mi355_csvm final : public pa::mi355::csvm {
  public:
    using pa::mi355::csvm::csvm;
    using pa::mi355::csvm::predict_values;
    using pa::mi355::csvm::solve_system_of_linear_equations;
};

template <typename T>
static void test_solve_trivial(pa::kernel_function_type kernel) {
    // GenericCSVM.solve_system_of_linear_equations_trivial (generic_csvm_tests.hpp:99-137)
    pa::detail::parameter<T> params;
    params.kernel_type = kernel;
    params.cost = T(2.0);
    if (kernel == pa::kernel_function_type::polynomial) {
        params.degree = 1;
        params.set_gamma(T(1.0));
        params.coef0 = T(0.0);
    }
    const T v = std::sqrt(T(1.0) - 1 / params.cost);
    const std::vector<std::vector<T>> A = { { v, 0, 0, 0 }, { 0, v, 0, 0 }, { 0, 0, v, 0 }, { 0, 0, 0, v } };
    const std::vector<T> rhs{ T(1.0), T(-1.0), T(1.0), T(-1.0) };
    const mock_mi355_csvm svm{ static_cast<pa::parameter>(params) };
    const auto [calculated_x, calculated_rho] = svm.solve_system_of_linear_equations(params, A, rhs, T(0.00001), A.front().size());
    EXPECT_TRUE(fp_vec_near(calculated_x, rhs));
    EXPECT_TRUE(std::abs(calculated_rho) < 8 * std::numeric_limits<T>::epsilon());
    EXPECT_TRUE(svm.last_cg_info().iterations >= 1 && svm.last_cg_info().iterations <= 4);
}

template <typename T>
static void test_predict_values(pa::kernel_function_type kernel) {
    // GenericCSVM.predict_values (generic_csvm_tests.hpp:149-195)
    pa::detail::parameter<T> params;
    params.kernel_type = kernel;
    params.cost = T(2.0);
    if (kernel == pa::kernel_function_type::polynomial) {
        params.degree = 1;
        params.set_gamma(T(1.0));
        params.coef0 = T(0.0);
    }
    const std::vector<std::vector<T>> support_vectors = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    const std::vector<T> weights{ T(1.0), T(-1.0), T(1.0), T(-1.0) };
    std::vector<T> w{};
    const std::vector<std::vector<T>> data{ { 1, 1, 1, 1 }, { 1, -1, 1, -1 } };
    const mock_mi355_csvm svm{ static_cast<pa::parameter>(params) };
    const std::vector<T> calculated = svm.predict_values(params, support_vectors, weights, T(0.0), w, data);
    EXPECT_TRUE(calculated.size() == data.size());
    EXPECT_TRUE(fp_vec_near(calculated, std::vector<T>{ T(0.0), T(4.0) }));
    if (kernel == pa::kernel_function_type::linear) {
        EXPECT_TRUE(w.size() == 4 && fp_vec_near(w, weights));
    } else {
        EXPECT_TRUE(w.empty());
    }
}

template <typename T>
static void test_fit_predict_score(pa::kernel_function_type kernel) {
    // two well separated blobs: a fitted model must classify its own training data (cf. GenericCSVM.predict / score,
    // generic_csvm_tests.hpp:197-247, which use LIBSVM-trained fixtures)
    std::mt19937 gen(7);
    std::normal_distribution<T> noise(T(0), T(0.3));
    const std::size_t N = 400, d = 10;
    std::vector<std::vector<T>> X(N, std::vector<T>(d));
    std::vector<T> y(N);
    for (std::size_t i = 0; i < N; ++i) {
        y[i] = (i % 2 == 0) ? T(1) : T(-1);
        for (std::size_t f = 0; f < d; ++f) X[i][f] = y[i] * T(0.5) + noise(gen);
    }
    pa::parameter params;
    params.kernel_type = kernel;
    const auto svm = pa::make_csvm(pa::backend_type::mi355, params);
    EXPECT_TRUE(svm->get_target_platform() == pa::target_platform::gpu_amd);
    auto m = svm->fit(X, y, T(1e-6));
    EXPECT_TRUE(m.alpha.size() == N);
    EXPECT_TRUE(std::abs(m.params.gamma - 1.0 / d) < 1e-15);  // default gamma = 1 / num_features (csvm.hpp:303-307)
    T sum = 0;
    for (const T a : m.alpha) sum += a;
    EXPECT_TRUE(std::abs(sum) < T(1e-3));  // alpha_N = -sum(alpha) (csvm.cpp:180)
    EXPECT_TRUE(svm->score(m, X, y) > T(0.99));
    const std::vector<T> labels = svm->predict(m, X);
    EXPECT_TRUE(labels.size() == N && (labels[0] == T(1) || labels[0] == T(-1)));
}

static void test_named_arguments_and_tracking(bool have_gpu) {
    // the named-parameter constructors (HIP/csvm.hpp:82-99, tests/backends/HIP/hip_csvm.cpp "construct_target_and_named_args")
    if (!have_gpu) {
        EXPECT_THROW_WHAT((pa::mi355::csvm{ pa::kernel_type = pa::kernel_function_type::rbf, pa::gamma = 0.25 }), pa::mi355::backend_exception, "no HIP capable devices were found");
        EXPECT_THROW_WHAT((pa::mi355::csvm{ pa::kernel_type = pa::kernel_function_type::rbf, pa::gamma = -0.25 }), pa::invalid_parameter_exception, "gamma must be greater than 0.0");
        return;
    }
    const pa::mi355::csvm svm{ pa::target_platform::gpu_amd, pa::kernel_type = pa::kernel_function_type::polynomial, pa::degree = 2, pa::gamma = 0.25, pa::coef0 = 1.5, pa::cost = 4.0,
                               pa::num_devices = 1 };
    const pa::parameter p = svm.get_params();
    EXPECT_TRUE(p.kernel_type == pa::kernel_function_type::polynomial && p.degree == 2 && p.gamma == 0.25 && !p.gamma_is_default && p.coef0 == 1.5 && p.cost == 4.0);
    EXPECT_TRUE(svm.get_num_devices() == 1);
    const auto via_factory = pa::make_csvm(pa::backend_type::mi355, pa::kernel_type = pa::kernel_function_type::rbf, pa::cost = 2.0);
    EXPECT_TRUE(via_factory->get_params().kernel_type == pa::kernel_function_type::rbf && via_factory->get_params().gamma_is_default && via_factory->get_params().cost == 2.0);
    EXPECT_THROW_WHAT((pa::mi355::csvm{ pa::num_devices = 4096 }), pa::mi355::backend_exception, "devices, but only");
    EXPECT_THROW_WHAT((pa::mi355::csvm{ pa::target_platform::gpu_nvidia, pa::cost = 2.0 }), pa::mi355::backend_exception, "Invalid target platform");
    // tracking entries of a solve in the reference's YAML layout (performance_tracker.cpp:139-190)
    const std::vector<std::vector<double>> X = { { 1, 2 }, { 3, 4 }, { 5, 7 }, { -1, 0 } };
    const std::vector<double> y = { 1, -1, 1, -1 };
    pa::mi355::csvm lin{ pa::kernel_type = pa::kernel_function_type::linear };
    (void) lin.fit(X, y, 1e-8);
    std::ostringstream yaml;
    lin.write_tracking_yaml(yaml);
    const std::string t = yaml.str();
    EXPECT_TRUE(t.rfind("---\n", 0) == 0 && t.find("backend:\n  backend: mi355\n") != std::string::npos && t.find("cg:\n  iterations: ") != std::string::npos
                && t.find("  target_residuum: ") != std::string::npos && t.find("  epsilon: 1e-08\n") != std::string::npos);
}

template <typename T>
static void test_multi_device_shards(pa::kernel_function_type kernel) {
    // the sharded solve behind ONE call (lssvm_mi355_solve_multi_*): three shards on device 0 against the plain single-device solve
    std::mt19937 gen(11);
    std::uniform_real_distribution<T> u(T(-1), T(1));
    const std::size_t N = 700, d = 24;
    std::vector<T> X(N * d), y(N);
    for (auto &v : X) v = u(gen);
    for (std::size_t i = 0; i < N; ++i) y[i] = (i % 2 == 0) ? T(1) : T(-1);
    const lssvm_params prm{ static_cast<int32_t>(kernel), 3, 1.0 / d, 0.0, 1.0 };
    std::vector<T> a1(N), a3(N);
    T rho1{}, rho3{};
    lssvm_cg_info i1{}, i3{};
    const int devs[3] = { 0, 0, 0 };
    int rc1, rc3;
    // the yardstick of an fp32 CG trajectory is its distance to the float64 solve (rounding noise is amplified from iteration to iteration)
    std::vector<double> Xd(X.begin(), X.end()), yd(y.begin(), y.end()), a64(N);
    double rho64{};
    const int rc64 = lssvm_mi355_solve_f64(&prm, Xd.data(), N, d, yd.data(), 1e-9, 60, a64.data(), &rho64, nullptr, nullptr);
    if constexpr (std::is_same_v<T, float>) {
        rc1 = lssvm_mi355_solve_f32(&prm, X.data(), N, d, y.data(), T(1e-5), 60, a1.data(), &rho1, &i1, nullptr);
        rc3 = lssvm_mi355_solve_multi_f32(&prm, X.data(), N, d, y.data(), T(1e-5), 60, a3.data(), &rho3, &i3, devs, 3, nullptr);
    } else {
        rc1 = lssvm_mi355_solve_f64(&prm, X.data(), N, d, y.data(), T(1e-9), 60, a1.data(), &rho1, &i1, nullptr);
        rc3 = lssvm_mi355_solve_multi_f64(&prm, X.data(), N, d, y.data(), T(1e-9), 60, a3.data(), &rho3, &i3, devs, 3, nullptr);
    }
    EXPECT_TRUE(rc1 == 0 && rc3 == 0 && rc64 == 0);
    if (rc3 != 0) std::printf("  %s\n", lssvm_mi355_last_error());
    EXPECT_TRUE(i1.devices_used == 1 && i3.devices_used == 3 && i3.local_devices == 3 && i3.exchange == 2);
    double err1 = 0, err3 = 0, scale = 0;
    for (std::size_t i = 0; i < N; ++i) {
        err1 = std::max(err1, std::abs(static_cast<double>(a1[i]) - a64[i]));
        err3 = std::max(err3, std::abs(static_cast<double>(a3[i]) - a64[i]));
        scale = std::max(scale, std::abs(a64[i]));
    }
    // both solves are converged to the stop criterion (eps = 1e-5 / 1e-9): they agree at that accuracy, not at rounding accuracy
    // (fp32: two CG trajectories of this system separate by a few 1e-3 of the solution's scale within a dozen iterations -- the sums of the sharded
    // solve are ordered differently; tests/test_gpu_parity.py holds both to the float64 oracle and the fp32 CPU oracle)
    const double tol = std::is_same_v<T, float> ? 3e-3 : 1e-6;
    EXPECT_TRUE(err3 <= 2 * err1 + tol * scale);
    if (!(err3 <= 2 * err1 + tol * scale)) {
        std::printf("  kernel %d, %s: err1 %.3e err3 %.3e scale %.3e, iterations %llu / %llu, gram mode %d / %d, rho %.6e / %.6e / %.6e\n", static_cast<int>(kernel),
                    std::is_same_v<T, float> ? "float" : "double", err1, err3, scale, static_cast<unsigned long long>(i1.iterations), static_cast<unsigned long long>(i3.iterations),
                    i1.gram_mode, i3.gram_mode, static_cast<double>(rho1), static_cast<double>(rho3), rho64);
    }
    EXPECT_TRUE(std::abs(static_cast<double>(rho3) - rho64) <= 2 * std::abs(static_cast<double>(rho1) - rho64) + 20 * tol * std::max(1.0, std::abs(rho64)));  // rho = -(y_N + QA_cost sum(x) - q.x) cancels
}

/* ABI 4: the tuning knobs belong to the csvm OBJECT.  Two objects with different gram_mode solve the same system at the same time from two threads, several times
 * over: every solve must report its own object's mode (0 = native v_mfma_f32, 1 = bf16x6, 2 = f16x3), the process defaults must stay untouched, and a third thread
 * that keeps flipping the process default must not leak into either (the boundary's semantics: include/plssvm/csvm.hpp:50-83 -- move-only objects, const virtuals,
 * no shared state beyond `verbosity`). */
static void test_two_objects_with_their_own_options_on_two_threads() {
    std::mt19937 gen(5);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const std::size_t N = 1500, d = 40;
    std::vector<std::vector<float>> X(N, std::vector<float>(d));
    std::vector<float> y(N);
    for (auto &row : X) for (auto &v : row) v = u(gen);
    for (std::size_t i = 0; i < N; ++i) y[i] = (i % 2 == 0) ? 1.0f : -1.0f;
    mock_mi355_csvm a{ pa::kernel_type = pa::kernel_function_type::rbf, pa::gamma = 0.05 };
    mock_mi355_csvm b{ pa::kernel_type = pa::kernel_function_type::rbf, pa::gamma = 0.05 };
    a.set_option("gram_mode", 1);
    b.set_option("gram_mode", 0);
    int64_t before = -1, after = -1;
    EXPECT_TRUE(lssvm_mi355_get_option("gram_mode", &before) == 0);
    EXPECT_TRUE(a.get_option("gram_mode") == 1 && b.get_option("gram_mode") == 0 && before == 3);
    pa::detail::parameter<float> prm{};
    prm.kernel_type = pa::kernel_function_type::rbf;
    prm.set_gamma(0.05f);
    std::atomic<int> wrong{ 0 };
    std::atomic<bool> stop{ false };
    std::vector<float> alpha_a, alpha_b;
    const auto run = [&](const mock_mi355_csvm &svm, int want, std::vector<float> &keep) {
        try {
            for (int rep = 0; rep < 6; ++rep) {
                auto res = svm.solve_system_of_linear_equations(prm, X, y, 1e-4f, 40);
                if (svm.last_cg_info().gram_mode != want) ++wrong;
                keep = std::move(res.first);
            }
        } catch (const std::exception &e) {
            std::printf("  thread for mode %d: %s\n", want, e.what());
            wrong += 100;
        }
    };
    std::thread flip([&] {  // the process default changes under both: neither may see it
        int64_t v = 2;
        while (!stop.load()) {
            (void) lssvm_mi355_set_option("gram_mode", v);
            v = v == 2 ? 3 : 2;
            std::this_thread::yield();
        }
        (void) lssvm_mi355_set_option("gram_mode", 3);
    });
    std::thread ta(run, std::cref(a), 1, std::ref(alpha_a)), tb(run, std::cref(b), 0, std::ref(alpha_b));
    ta.join();
    tb.join();
    stop.store(true);
    flip.join();
    EXPECT_TRUE(wrong.load() == 0);
    EXPECT_TRUE(lssvm_mi355_get_option("gram_mode", &after) == 0 && after == 3);
    // the two paths solve the same system: they agree as two converged fp32 solves do
    double diff = 0, scale = 0;
    for (std::size_t i = 0; i < alpha_a.size() && i < alpha_b.size(); ++i) {
        diff = std::max(diff, std::abs(static_cast<double>(alpha_a[i]) - alpha_b[i]));
        scale = std::max(scale, std::abs(static_cast<double>(alpha_a[i])));
    }
    EXPECT_TRUE(alpha_a.size() == N && alpha_b.size() == N && diff <= 5e-3 * scale);
    // an object that never set an option follows the process defaults; option errors are the library's
    mock_mi355_csvm c{ pa::kernel_type = pa::kernel_function_type::rbf, pa::gamma = 0.05 };
    (void) c.solve_system_of_linear_equations(prm, X, y, 1e-4f, 5);
    EXPECT_TRUE(c.last_cg_info().gram_mode == 2 || c.last_cg_info().gram_mode == 1);
    EXPECT_THROW_WHAT(c.set_option("no_such_option", 1), pa::invalid_parameter_exception, "unknown option");
    EXPECT_THROW_WHAT(c.set_option("gram_mode", 9), pa::invalid_parameter_exception, "gram_mode must be");
}

static void test_factory_and_exceptions(bool have_gpu) {
    // tests/csvm_factory.cpp:61-212
    EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::cuda), pa::unsupported_backend_exception, "No cuda backend available!");
    EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::openmp), pa::unsupported_backend_exception, "No openmp backend available!");
    EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::sycl), pa::unsupported_backend_exception, "No sycl backend available!");
    pa::parameter bad;
    bad.kernel_type = pa::kernel_function_type::rbf;
    bad.set_gamma(-1.0);
    EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::mi355, bad), pa::invalid_parameter_exception, "gamma must be greater than 0.0");  // csvm.hpp:384
    if (!have_gpu) {
        // no device: construction must fail loudly (csvm.hip.cpp:70-72), there is no CPU fallback
        EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::mi355), pa::mi355::backend_exception, "no HIP capable devices were found");
        EXPECT_THROW_WHAT((void) pa::make_csvm(), pa::mi355::backend_exception, "no HIP capable devices were found");
        return;
    }
    EXPECT_THROW_WHAT((void) pa::make_csvm(pa::backend_type::mi355, pa::target_platform::cpu), pa::mi355::backend_exception, "Invalid target platform");  // csvm.hip.cpp:49-51
    const auto a = pa::make_csvm();
    const auto b = pa::make_csvm(pa::backend_type::hip, pa::target_platform::gpu_amd, pa::parameter{});
    EXPECT_TRUE(dynamic_cast<pa::mi355::csvm *>(a.get()) != nullptr && dynamic_cast<pa::mi355::csvm *>(b.get()) != nullptr);
    EXPECT_TRUE(a->get_target_platform() == pa::target_platform::gpu_amd);
    // fit argument validation (tests/csvm.cpp:196-340)
    const std::vector<std::vector<double>> X = { { 1, 2 }, { 3, 4 }, { 5, 6 } };
    const std::vector<double> y = { 1, -1, 1 };
    EXPECT_THROW_WHAT((void) a->fit(X, y, 0.0), pa::invalid_parameter_exception, "epsilon must be less than 0.0");
    EXPECT_THROW_WHAT((void) a->fit(X, std::vector<double>{}), pa::invalid_parameter_exception, "No labels given for training");
    const std::vector<std::vector<double>> ragged = { { 1, 2 }, { 3 }, { 5, 6 } };
    EXPECT_THROW_WHAT((void) a->fit(ragged, y), pa::invalid_parameter_exception, "same number of features");
    auto m = a->fit(X, y);
    const std::vector<std::vector<double>> wrong = { { 1, 2, 3 } };
    EXPECT_THROW_WHAT((void) a->predict(m, wrong), pa::invalid_parameter_exception, "must match the number of features per support vector");
}

int main(int argc, char **argv) {
    const bool no_gpu = argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0;
    const bool have_gpu = lssvm_mi355_device_count() > 0;
    if (no_gpu && have_gpu) std::printf("note: --no-gpu given but a device is visible; running the no-GPU subset anyway\n");
    test_factory_and_exceptions(have_gpu && !no_gpu);
    test_named_arguments_and_tracking(have_gpu && !no_gpu);
    if (!no_gpu) {
        if (!have_gpu) {
            std::printf("no HIP device visible: run with --no-gpu for the CPU-only subset\n");
            return 99;
        }
        for (const auto k : { pa::kernel_function_type::linear, pa::kernel_function_type::polynomial }) {  // rbf is skipped by the reference (generic_csvm_tests.hpp:110-112)
            test_solve_trivial<float>(k);
            test_solve_trivial<double>(k);
            test_predict_values<float>(k);
            test_predict_values<double>(k);
        }
        for (const auto k : { pa::kernel_function_type::linear, pa::kernel_function_type::polynomial, pa::kernel_function_type::rbf }) {
            test_fit_predict_score<float>(k);
            test_fit_predict_score<double>(k);
            test_multi_device_shards<float>(k);
            test_multi_device_shards<double>(k);
        }
        test_two_objects_with_their_own_options_on_two_threads();
    }
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed;
}
