"""The product's BVH builder/flattener (librthost, index-based, threaded) must be leaf-for-leaf identical
to the oracle's literal restatement of constructBVHWithLimit (app/Hittable/BVH.hs:29-48)."""
import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import _ffi as F
from raytracer_amd import api, scenes


def assert_same_tree(spec, n_threads=1):
    k, d, m = spec.all_objects()
    b = api.HostBvh(k, d, m, spec.leaf_threshold, spec.bvh_max_depth, n_threads)
    o = O.OracleScene(spec)
    a, e = b.arrays(), o.bvh_export()
    info = o.bvh_info()
    assert (b.n_nodes, b.n_leaves, b.depth, b.n_prims) == (info["n_nodes"], info["n_leaves"], info["depth"], info["n_prims"])
    for key in a:
        assert np.array_equal(a[key], e[key]), key
    return b


@pytest.mark.parametrize("mk", [scenes.c1_testing, scenes.c2_spheres, scenes.c3_blob988])
def test_small_scenes_identical(mk):
    assert_same_tree(mk())


def test_c4_63k_identical_serial_and_threaded():
    spec = scenes.c4_blob63k()
    b = assert_same_tree(spec, 1)
    assert b.n_prims == 63432 and b.n_leaves == 16384 and b.depth == 14      # 16,384 leaves of 3-4 tris
    assert_same_tree(spec, 8)


def test_depth_limit_variant_and_leaf_threshold():
    spec = scenes.c4_blob63k(bvh_max_depth=10)                               # obli.json:41 as shipped: 1,024 leaves x ~62
    b = assert_same_tree(spec)
    assert b.n_leaves == 1024 and b.depth == 10
    spec = scenes.c3_blob988(leaf_threshold=1)
    assert_same_tree(spec)
    spec = scenes.c3_blob988(bvh_max_depth=0)                                # depth >= maxDepth at the root: one leaf
    b = assert_same_tree(spec)
    assert b.n_nodes == 1


def test_stable_sort_keeps_input_order_for_equal_centroids():
    # 6 identical spheres: every centroid key ties; a stable sort must keep allObjects order in every leaf
    spec = scenes.c2_spheres()
    spec.config_objects = [("sphere", [0, 0, 0], 1.0, None)] * 6
    spec.leaf_threshold = 1
    b = assert_same_tree(spec)
    assert b.arrays()["prim_id"].tolist() == [0, 1, 2, 3, 4, 5]


def test_split_axis_rule():
    # extent x == extent y > z: x is NOT strictly largest -> y wins over z (BVH.hs:69-71)
    spec = scenes.c2_spheres()
    spec.config_objects = [("sphere", [0, 0, 0], 0.1, None), ("sphere", [1, 1, 0], 0.1, None)]
    spec.leaf_threshold = 1
    b = assert_same_tree(spec)
    a = b.arrays()
    assert a["prim_id"].tolist() == [0, 1]


def test_large_random_scene_threaded():
    rng = np.random.default_rng(5)
    n = 200_000
    c = rng.uniform(-50, 50, (n, 3))
    tri = np.concatenate([c, c + rng.uniform(-0.3, 0.3, (n, 3)), c + rng.uniform(-0.3, 0.3, (n, 3))], axis=1)
    kind = np.zeros(n, np.int32); mat = np.zeros(n, np.int32)
    b1 = api.HostBvh(kind, tri, mat, 4, 32, 1)
    b8 = api.HostBvh(kind, tri, mat, 4, 32, 8)
    a1, a8 = b1.arrays(), b8.arrays()
    for key in a1:
        assert np.array_equal(a1[key], a8[key]), key
    # every primitive exactly once, every leaf within the threshold
    assert sorted(a1["prim_id"].tolist()) == list(range(n))
    leaves = a1["node_child"][a1["node_child"][:, 0] < 0]
    assert leaves[:, 1].max() <= 4 and leaves[:, 1].sum() == n


def test_error_cases():
    k = np.zeros(0, np.int32); d = np.zeros((0, 9)); m = np.zeros(0, np.int32)
    with pytest.raises(F.RtError) as e:
        api.HostBvh(k, d, m, 4, 32)
    assert e.value.code == F.RT_ERR_EMPTY_SCENE                              # BVH.hs:102
    with pytest.raises(F.RtError):
        api.HostBvh(np.zeros(1, np.int32), np.zeros((1, 9)), np.zeros(1, np.int32), 0, 32)   # leafThreshold 0 never terminates
    with pytest.raises(F.RtError):
        api.HostBvh(np.full(1, 7, np.int32), np.zeros((1, 9)), np.zeros(1, np.int32), 4, 32)


def test_default_camera_matches_oracle():
    for spec in (scenes.c1_testing(), scenes.c4_blob63k(width=64, height=36)):
        a = api.default_camera(spec.camera, spec.width / spec.height)
        b = O.default_camera(spec.camera, spec.width / spec.height)
        assert bytes(a) == bytes(b)
