"""Bit source and BER helpers (host-side mirror of dev/py/util.py:25-41)."""
import numpy as np


def bit_source(nbits, frame_size, msg_length):
    """Uniform random label bits [msg_length frames, frame_size data cells, nbits] from ``np.random``."""
    return np.random.randint(0, 2, (int(msg_length), int(frame_size), int(nbits)))


def ber_calc(conf_matrix):
    """(c01 + c10) / total of a 2x2 confusion matrix (row = label, column = decision)."""
    conf_matrix = np.asarray(conf_matrix)
    if conf_matrix.shape != (2, 2):
        raise AssertionError("2x2 confusion matrix expected")
    return float(conf_matrix[0][1] + conf_matrix[1][0]) / float(np.sum(conf_matrix))
