"""Model-half parity against the reference's own TensorFlow numbers (tests/golden/tf_golden.npz, written by
tools/make_tf_golden.py where TensorFlow exists).  Absent file => skipped: the model half then stays "parity unpinned"
(DESIGN.md §2 row c).  The consumer code itself is exercised on a file of the same schema synthesized from the oracle."""
import os

import numpy as np
import pytest

import tf_golden as tg


def test_consumers_work_on_a_synthesized_file(tmp_path):
    path = tg.synthesize(str(tmp_path / "tf_like.npz"), cases=("mixednet_default", "inception_default"), mode="keras_last_axis")
    z = np.load(path)
    for case in z["cases"]:
        assert tg.check_oracle(z, str(case)) == ["keras_last_axis"]


@pytest.mark.skipif(not os.path.isfile(tg.GOLDEN), reason="tests/golden/tf_golden.npz absent: run tools/make_tf_golden.py where TensorFlow is installed")
def test_oracle_matches_the_reference_tensorflow_model():
    from microwakeword_amd import model
    z = np.load(tg.GOLDEN)
    for case in z["cases"]:
        modes = tg.check_oracle(z, str(case))
        # the package's reading of the [B,B] sample weight matrix has to be the one Keras follows
        assert model.MATRIX_WEIGHT_BROADCAST in modes, (
            "the reference's weighted loss follows %s; set microwakeword_amd.model.MATRIX_WEIGHT_BROADCAST accordingly" % modes)
