"""Validation scores with the names and definitions of the reference's
``src/matrix_factorization/metrics.py`` (``classification_scores`` ``:30-57``,
``regression_scores`` ``:60-85``); sklearn on host arrays, exactly as there."""
import numpy as np
from sklearn.metrics import accuracy_score, roc_auc_score, mean_absolute_error, mean_squared_error


def round_probabilities(probabilities, threshold):
    labels = np.zeros_like(probabilities, dtype=np.uint8)
    labels[probabilities >= threshold] = 1
    return labels


def classification_scores(y_true, y_pred, threshold=0.5):
    y_pred_labels = round_probabilities(y_pred, threshold=threshold)
    return {'accuracy': accuracy_score(y_true, y_pred_labels), 'roc_auc': roc_auc_score(y_true, y_pred)}


def regression_scores(y_true, y_pred):
    return {'mean_absolute_error': mean_absolute_error(y_true, y_pred), 'mean_squared_error': mean_squared_error(y_true, y_pred)}
