"""nsr/configs.json is exactly what the reference's YAMLs resolve to (build container only)."""
import os

import pytest


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree only exists in the build container")
def test_configs_match_reference_yamls():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_configs
    import nsr
    want = gen_configs.resolved()
    assert sorted(want) == nsr.configs.names()
    for k, v in want.items():
        assert nsr.configs.get(k) == v, k


def test_hot_path_shapes():
    import nsr
    c2, c5 = nsr.configs.get("nerf-blender"), nsr.configs.get("neuralangelo")
    assert c2["geometry"]["xyz_encoding_config"]["log2_hashmap_size"] == 19 and c2["max_train_num_rays"] == 8192
    assert c5["geometry"]["grad_type"] == "finite_difference" and c5["texture"]["mlp_network_config"]["otype"] == "VanillaMLP"
