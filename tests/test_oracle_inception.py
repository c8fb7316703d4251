"""oracle/inception_oracle.py against what the reference's own test holds for this path (end-point shapes and the
variable total, nets/inception_v3_test.py:87-104, :112-120) and against an independent torch statement."""
import numpy as np
import torch

from oracle import inception_oracle as io
from tests import _torch_ref as tr

REF_SHAPES = {'Conv2d_1a_3x3': (149, 149, 32), 'Conv2d_2a_3x3': (147, 147, 32), 'Conv2d_2b_3x3': (147, 147, 64),
              'MaxPool_3a_3x3': (73, 73, 64), 'Conv2d_3b_1x1': (73, 73, 80), 'Conv2d_4a_3x3': (71, 71, 192),
              'MaxPool_5a_3x3': (35, 35, 192), 'Mixed_5b': (35, 35, 256), 'Mixed_5c': (35, 35, 288), 'Mixed_5d': (35, 35, 288),
              'Mixed_6a': (17, 17, 768), 'Mixed_6b': (17, 17, 768), 'Mixed_6c': (17, 17, 768), 'Mixed_6d': (17, 17, 768),
              'Mixed_6e': (17, 17, 768), 'Mixed_7a': (8, 8, 1280), 'Mixed_7b': (8, 8, 2048), 'Mixed_7c': (8, 8, 2048)}


def test_endpoint_shapes_and_variable_total_match_the_reference_test():
    shapes = io.endpoint_shapes(299, 299, batch=5)
    assert list(shapes) == list(REF_SHAPES)                                  # inception_v3_test.py:55-59 (order)
    for k, v in REF_SHAPES.items():
        assert shapes[k] == (5,) + v, k                                       # :87-104
    assert sum(int(np.prod(s)) for _, s in io.param_specs()) == 21802784     # :120
    assert io.endpoint_shapes(125, 125)["Mixed_7c"] == (1, 2, 2, 2048)        # the sampler's 125x125 frames


def test_oracle_matches_torch_statement():
    p = io.init_params(0)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (2, 125, 125, 3))
    feat = io.forward(p, x)["Mixed_7c"]
    tp = {k: torch.tensor(v) for k, v in p.items()}
    tfeat = tr.inception_v3_mixed7c(tp, torch.tensor(x)).numpy()
    assert feat.shape == (2, 2, 2, 2048) and feat.min() >= 0 and feat.max() > 0
    np.testing.assert_allclose(feat, tfeat, rtol=1e-9, atol=1e-10)


def test_avg_pool_counts_only_taps_inside_the_image():
    net = io.Net({})
    x = np.ones((1, 3, 3, 1))
    np.testing.assert_allclose(net.avg_pool(x), np.ones((1, 3, 3, 1)))       # a constant image stays constant at the border
    x = np.arange(9.0).reshape(1, 3, 3, 1)
    assert net.avg_pool(x)[0, 0, 0, 0] == (0 + 1 + 3 + 4) / 4
