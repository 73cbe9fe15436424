"""``plssvm-train`` / ``plssvm-predict`` / ``plssvm-scale`` compatible command lines (src/main_train.cpp:24-70,
src/main_predict.cpp:29-100, src/main_scale.cpp:25-85, src/plssvm/detail/cmd/parser_train.cpp:41-73, parser_predict.cpp:44-60,
parser_scale.cpp:41-135) on top of the MI355X backend.

    python -m plssvm_amd.train   [-t -d -g -r -c -e -i -b -p --use_strings_as_labels --use_float_as_real_type --verbosity -q] training_set_file [model_file]
    python -m plssvm_amd.predict [-b -p --use_strings_as_labels --use_float_as_real_type --verbosity -q] test_file model_file [output_file]
    python -m plssvm_amd.scale   [-l -u -f -s -r --use_strings_as_labels --use_float_as_real_type --verbosity -q] input_file [scaled_file]
"""

from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from .csvm import BackendType, TargetPlatform, make_csvm
from .data_set import DataSet, Scaling
from .exceptions import PlssvmError
from .model import Model
from .parameter import Parameter

VERBOSITY = ("full", "timing", "libsvm", "quiet")

# the phases of the last train_main / predict_main call of this process, in seconds: what the reference's tracker records as data_set_read / cg / model_write times
# (main_train.cpp:24-70); bench.py prints them as its `e2e` block
LAST_TIMINGS: dict = {}


def _common(ap):
    ap.add_argument("-b", "--backend", default="automatic", help="choose the backend: automatic|mi355|hip (the reference's other backends are not built here)")
    ap.add_argument("-p", "--target_platform", default="automatic", help="choose the target platform: automatic|gpu_amd")
    ap.add_argument("--use_strings_as_labels", action="store_true", help="use strings as labels instead of plane numbers")
    ap.add_argument("--use_float_as_real_type", action="store_true", help="use floats as real types instead of doubles")
    ap.add_argument("--verbosity", choices=VERBOSITY, default=None, help="choose the level of verbosity: full|timing|libsvm|quiet (default: full)")
    ap.add_argument("-q", "--quiet", action="store_true", help="quiet mode (no outputs regardless the provided verbosity level!)")
    ap.add_argument("-v", "--version", action="store_true", help="print version information")


def _verbosity(args):
    if args.quiet:
        if args.verbosity not in (None, "quiet"):
            print(f'WARNING: explicitly set the -q/--quiet flag, but the provided verbosity level isn\'t "quiet"; setting --verbosity={args.verbosity} to --verbosity=quiet',
                  file=sys.stderr)
        return "quiet"
    return args.verbosity or "full"


def _log(verb, levels, text):
    """detail::log(verbosity_level, ...) (logger.hpp:109-123): printed iff the active level is one of `levels`."""
    if verb != "quiet" and verb in levels:
        print(text, flush=True)


def train_main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="plssvm-train", description="LS-SVM with multiple (GPU-)backends")
    ap.add_argument("-t", "--kernel_type", default="0", help="set type of kernel function. 0 -- linear: u'*v, 1 -- polynomial: (gamma*u'*v + coef0)^degree, 2 -- radial basis function: exp(-gamma*|u-v|^2)")
    ap.add_argument("-d", "--degree", type=int, default=3, help="set degree in kernel function")
    ap.add_argument("-g", "--gamma", type=float, default=None, help="set gamma in kernel function (default: 1 / num_features)")
    ap.add_argument("-r", "--coef0", type=float, default=0.0, help="set coef0 in kernel function")
    ap.add_argument("-c", "--cost", type=float, default=1.0, help="set the parameter C")
    ap.add_argument("-e", "--epsilon", type=float, default=0.001, help="set the tolerance of termination criterion")
    ap.add_argument("-i", "--max_iter", type=int, default=None, help="set the maximum number of CG iterations (default: num_features)")
    ap.add_argument("--performance_tracking", default=None, metavar="FILE",
                    help="the output YAML file where the performance tracking results are written to (appended); '-' dumps them to stderr")  # parser_train.cpp
    _common(ap)
    ap.add_argument("input", nargs="?", help="training_set_file")
    ap.add_argument("model", nargs="?", help="model_file")
    args = ap.parse_args(argv)
    if args.version:
        print("plssvm_amd (MI355X-native LS-SVM CG backend), C ABI version 2")
        return 0
    if args.input is None:
        print("Error missing input file!", file=sys.stderr)
        ap.print_help()
        return 1
    if args.max_iter is not None and args.max_iter <= 0:
        print(f"max_iter must be greater than 0, but is {args.max_iter}!", file=sys.stderr)  # parser_train.cpp:130-133
        return 1
    verb = _verbosity(args)
    model_file = args.model if args.model is not None else os.path.basename(args.input) + ".model"  # parser_train.cpp:220-226
    try:
        t0 = time.perf_counter()
        real_type = np.float32 if args.use_float_as_real_type else np.float64
        params = Parameter(kernel_type=args.kernel_type, degree=args.degree, gamma=args.gamma, coef0=args.coef0, cost=args.cost)
        _log(verb, ("full",), f"\ntask: training\nkernel_type: {params.kernel_type}\ncost: {params.cost}\nepsilon: {args.epsilon}\n"
                              f"real_type: {np.dtype(real_type).name}\ninput file (data set): '{args.input}'\noutput file (model): '{model_file}'\n")
        t_r = time.perf_counter()
        data = DataSet(filename=args.input, real_type=real_type, label_type=str if args.use_strings_as_labels else float)
        t_r = time.perf_counter() - t_r
        _log(verb, ("full", "timing"), f"Read {data.num_data_points()} data points with {data.num_features()} features using the libsvm parser from file '{args.input}'.")
        svm = make_csvm(args.backend, TargetPlatform(args.target_platform), params)
        _log(verb, ("full",), f"\nUsing MI355 as backend.\nFound {svm.num_devices} HIP device(s).\n")
        model = svm.fit(data, epsilon=args.epsilon, max_iter=args.max_iter)
        info = svm.last_cg_info
        _log(verb, ("full", "timing"), f"Finished after {info['iterations']}/{info['max_iterations']} iterations with a residuum of {info['residuum']} "
                                      f"(target: {info['target_residuum']}) and an average iteration time of {info['avg_iteration_ms']:.3f}ms.")
        _log(verb, ("full", "libsvm"), f"optimization finished, #iter = {info['iterations']}")  # csvm.cpp:175-176
        _log(verb, ("full", "timing"), f"Solved minimization problem (r = b - Ax) using the Conjugate Gradient (CG) methode in {info['total_runtime_ms']:.0f}ms.\n")
        t_w = time.perf_counter()
        model.save(model_file)
        t_w = time.perf_counter() - t_w
        _log(verb, ("full", "timing"), f"Write {model.num_support_vectors()} support vectors with {model.num_features()} features to the libsvm model file '{model_file}'.")
        LAST_TIMINGS.clear()
        LAST_TIMINGS.update({"task": "train", "read_s": t_r, "setup_ms": float(info.get("setup_ms", 0.0)), "solve_s": float(info.get("total_ms", 0.0)) * 1e-3,
                             "fit_s": float(info["total_runtime_ms"]) * 1e-3, "write_s": t_w, "total_s": time.perf_counter() - t0, "iterations": int(info["iterations"]),
                             "num_data_points": data.num_data_points(), "num_features": data.num_features(), "model_bytes": os.path.getsize(model_file)})
        if args.performance_tracking is not None:
            from .performance_tracker import PerformanceTracker

            tr = PerformanceTracker()
            tr.add_parameter(params, np.dtype(real_type).name)
            tr.add_backend(info.get("devices_used", 1))
            tr.add_cg_info(info)
            tr.add("data_set_read", "num_data_points", data.num_data_points())
            tr.add("data_set_read", "num_features", data.num_features())
            tr.add("data_set_read", "filename", args.input)
            tr.add("data_set_read", "time", f"{t_r * 1e3:.0f}ms")
            tr.add("model_write", "num_support_vectors", model.num_support_vectors())
            tr.add("model_write", "rho", float(model.rho))  # model.hpp:221
            tr.add("model_write", "filename", model_file)
            tr.add("model_write", "time", f"{t_w * 1e3:.0f}ms")
            tr.add("", "total_time", f"{(time.perf_counter() - t0) * 1e3:.0f}ms")  # main_train.cpp:57
            tr.save(None if args.performance_tracking == "-" else args.performance_tracking)
        _log(verb, ("full",), f"\nTotal runtime: {(time.perf_counter() - t0) * 1e3:.0f}ms")
    except PlssvmError as e:
        print(f"{e}\nException type: {type(e).__name__}", file=sys.stderr)
        return 1
    return 0


def predict_main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="plssvm-predict", description="LS-SVM with multiple (GPU-)backends")
    _common(ap)
    ap.add_argument("test", nargs="?", help="test_file")
    ap.add_argument("model", nargs="?", help="model_file")
    ap.add_argument("output", nargs="?", help="output_file")
    args = ap.parse_args(argv)
    if args.version:
        print("plssvm_amd (MI355X-native LS-SVM CG backend), C ABI version 2")
        return 0
    if args.test is None:
        print("Error missing test file!", file=sys.stderr)
        return 1
    if args.model is None:
        print("Error missing model file!", file=sys.stderr)
        return 1
    verb = _verbosity(args)
    out_file = args.output if args.output is not None else os.path.basename(args.test) + ".predict"  # parser_predict.cpp:150-156
    try:
        t0 = time.perf_counter()
        real_type = np.float32 if args.use_float_as_real_type else np.float64
        label_type = str if args.use_strings_as_labels else float
        t_r = time.perf_counter()
        data = _test_data(args.test, real_type, label_type)  # (the file is parsed ONCE: until round 6 a first pass looked at its labels only)
        t_r = time.perf_counter() - t_r
        t_m = time.perf_counter()
        model = Model.load(args.model, real_type=real_type, label_type=label_type)
        t_m = time.perf_counter() - t_m
        svm = make_csvm(args.backend, TargetPlatform(args.target_platform))
        t_p = time.perf_counter()
        predicted = svm.predict(model, data)
        t_p = time.perf_counter() - t_p
        t_w = time.perf_counter()
        with open(out_file, "w") as f:
            f.write("\n".join(_fmt(p) for p in predicted))
        t_w = time.perf_counter() - t_w
        LAST_TIMINGS.clear()
        LAST_TIMINGS.update({"task": "predict", "read_s": t_r, "model_read_s": t_m, "predict_s": t_p, "predict_phases": dict(getattr(svm, "last_predict_phases", {})), "write_s": t_w,
                             "num_data_points": data.num_data_points(), "num_support_vectors": model.num_support_vectors()})
        _log(verb, ("full", "timing"), f"Write {len(predicted)} predictions to the file '{out_file}'.")
        if data.has_labels():
            correct = sum(1 for p, c in zip(predicted, data.labels()) if p == c)
            # main_predict.cpp:75-79: the ratio in the real type, printed with {fmt}'s {} (shortest round trip: 100, 99.2)
            acc = real_type(correct) / real_type(len(predicted)) * real_type(100)
            acc_text = str(int(acc)) if float(acc).is_integer() else str(acc)
            _log(verb, ("full", "libsvm"), f"Accuracy = {acc_text}% ({correct}/{len(predicted)}) (classification)")
        _log(verb, ("full", "timing"), f"\nTotal runtime: {(time.perf_counter() - t0) * 1e3:.0f}ms")
    except PlssvmError as e:
        print(f"{e}\nException type: {type(e).__name__}", file=sys.stderr)
        return 1
    return 0


def scale_main(argv=None) -> int:
    """``plssvm-scale``: scale every feature of a LIBSVM file to [lower, upper] with factors computed from the file (optionally saved,
    -s) or restored from an earlier run (-r: the way a test set gets the scaling of its training set)."""
    ap = argparse.ArgumentParser(prog="plssvm-scale", description="LS-SVM with multiple (GPU-)backends")
    ap.add_argument("-l", "--lower", type=float, default=None, help="lower is the lowest (minimal) value allowed in each dimension (default: -1)")
    ap.add_argument("-u", "--upper", type=float, default=None, help="upper is the highest (maximal) value allowed in each dimension (default: 1)")
    ap.add_argument("-f", "--format", default="libsvm", help="the file format to output the scaled data set to: libsvm|arff")
    ap.add_argument("-s", "--save_filename", default=None, help="the file to which the scaling factors should be saved")
    ap.add_argument("-r", "--restore_filename", default=None, help="the file from which previous scaling factors should be loaded")
    ap.add_argument("--use_strings_as_labels", action="store_true", help="use strings as labels instead of plane numbers")
    ap.add_argument("--use_float_as_real_type", action="store_true", help="use floats as real types instead of doubles")
    ap.add_argument("--verbosity", choices=VERBOSITY, default=None, help="choose the level of verbosity: full|timing|libsvm|quiet (default: full)")
    ap.add_argument("-q", "--quiet", action="store_true", help="quiet mode (no outputs regardless the provided verbosity level!)")
    ap.add_argument("-v", "--version", action="store_true", help="print version information")
    ap.add_argument("input", nargs="?", help="input_file")
    ap.add_argument("scaled", nargs="?", help="scaled_file")
    args = ap.parse_args(argv)
    if args.version:
        print("plssvm_amd (MI355X-native LS-SVM CG backend), C ABI version 2")
        return 0
    lower = -1.0 if args.lower is None else args.lower
    upper = 1.0 if args.upper is None else args.upper
    if lower >= upper:  # parser_scale.cpp:89-93
        print(f"Error invalid scaling range [lower, upper] with [{_fmt(lower)}, {_fmt(upper)}]!", file=sys.stderr)
        return 1
    if args.format.lower() not in ("libsvm", "arff"):
        print(f"Error: the output format must be libsvm or arff, but \"{args.format}\" was given!", file=sys.stderr)
        return 1
    if args.input is None:
        print("Error missing input file!", file=sys.stderr)
        return 1
    if args.save_filename is not None and args.restore_filename is not None:
        print("Error cannot use -s (--save_filename) and -r (--restore_filename) simultaneously!", file=sys.stderr)
        return 1
    if args.restore_filename is not None and (args.lower is not None or args.upper is not None):
        print("Warning: provided -l (--lower) and/or -u (--upper) together with -r (--restore_filename); ignoring -l/-u", file=sys.stderr)
    verb = _verbosity(args)
    try:
        t0 = time.perf_counter()
        real_type = np.float32 if args.use_float_as_real_type else np.float64
        label_type = str if args.use_strings_as_labels else float
        scaling = Scaling(filename=args.restore_filename, real_type=real_type) if args.restore_filename is not None else Scaling(real_type(lower), real_type(upper))
        # (a name ending in .arff is read with the ARFF parser; labelled data: exactly two classes, as in the reference's data_set)
        data = DataSet(filename=args.input, real_type=real_type, label_type=label_type, scaling=scaling)
        _log(verb, ("full", "timing"), f"Scaled the data set to the range [{_fmt(scaling.lower)}, {_fmt(scaling.upper)}].")
        if args.scaled is not None:
            data.save(args.scaled, file_format=args.format)
        else:  # main_scale.cpp:41-62: the scaled data goes to stdout, "{label} {index}:{value:.10e} ..." per point
            print()
            Xs = data.data()
            for i in range(Xs.shape[0]):
                head = f"{_fmt(data.labels()[i])} " if data.has_labels() else ""
                print(head + "".join(f"{j + 1}:{float(Xs[i, j]):.10e} " for j in range(Xs.shape[1]) if Xs[i, j] != 0))
        if args.save_filename is not None:
            data.scaling_factors().save(args.save_filename)
        _log(verb, ("full", "timing"), f"\nTotal runtime: {(time.perf_counter() - t0) * 1e3:.0f}ms")
    except PlssvmError as e:
        print(f"{e}\nException type: {type(e).__name__}", file=sys.stderr)
        return 1
    return 0


def _fmt(label):
    if isinstance(label, str):
        return label
    f = float(label)
    return str(int(f)) if f.is_integer() else repr(f)


def _parse_by_extension(filename, dtype, label_type):
    """the test file through the parser its name asks for, like data_set's constructor (data_set.hpp:494-505): *.arff -> ARFF, else LIBSVM"""
    if str(filename).endswith(".arff"):
        from .io_arff import parse_arff_data
        return parse_arff_data(filename, dtype=dtype, label_type=label_type)
    from .io_libsvm import parse_libsvm_data
    return parse_libsvm_data(filename, dtype=dtype, label_type=label_type)


def _test_data(filename, real_type, label_type=float):
    """The test file of plssvm-predict: a labelled data set where it holds exactly two classes; without labels, or with a single class, it is usable for
    prediction only (labels kept for the accuracy line, parsed with the label type the command line asked for: --use_strings_as_labels must not turn "cat" into
    a float conversion error)."""
    X, labels = _parse_by_extension(filename, real_type, label_type)
    if labels is not None and len(set(labels)) == 2:
        return DataSet(X, labels, real_type=real_type, label_type=label_type)
    ds = DataSet(X, None, real_type=real_type)
    ds._labels = labels
    return ds
