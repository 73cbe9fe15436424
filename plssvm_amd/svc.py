"""sklearn-like ``SVC`` on top of CSVM (the reference's bindings/Python/sklearn.cpp): fit(X, y) / predict / score /
decision_function with the LS-SVM solver; only the parameters the LS-SVM path has are honoured."""

from __future__ import annotations

import numpy as np

from .csvm import make_csvm
from .data_set import DataSet
from .exceptions import InvalidParameterError
from .parameter import Parameter

__all__ = ["SVC"]


class SVC:
    def __init__(self, C=1.0, kernel="rbf", degree=3, gamma="scale_features", coef0=0.0, tol=1e-3, max_iter=-1, real_type=np.float64):
        self.C, self.kernel, self.degree, self.gamma, self.coef0, self.tol, self.max_iter, self.real_type = C, kernel, degree, gamma, coef0, tol, max_iter, real_type
        self.model_ = None
        self._svm = None

    def _params(self):
        if self.kernel not in ("linear", "poly", "polynomial", "rbf"):
            raise InvalidParameterError(f'The kernel "{self.kernel}" is not supported; use linear, poly or rbf')
        gamma = None if isinstance(self.gamma, str) else float(self.gamma)  # any string = the PLSSVM default 1 / n_features
        return Parameter(kernel_type=self.kernel, degree=self.degree, gamma=gamma, coef0=self.coef0, cost=self.C)

    def fit(self, X, y):
        self._svm = make_csvm(params=self._params())
        data = DataSet(X, list(np.asarray(y).tolist()), real_type=self.real_type)
        self.classes_ = np.array(data.different_labels())
        self.model_ = self._svm.fit(data, epsilon=self.tol, max_iter=None if self.max_iter is None or self.max_iter < 0 else self.max_iter)
        self.n_iter_ = int(self._svm.last_cg_info["iterations"])
        return self

    def decision_function(self, X):
        m = self.model_
        values, w = self._svm.predict_values(m.params, m.support_vectors(), m.alpha, float(m.rho), m.w, np.asarray(X, dtype=self.real_type))
        if w is not None:
            m.w = w
        return values

    def predict(self, X):
        return np.where(self.decision_function(X) > 0, self.classes_[1], self.classes_[0])

    def score(self, X, y):
        return float(np.mean(self.predict(X) == np.asarray(y)))
