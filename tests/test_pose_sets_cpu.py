"""CPU: the pose sets and visiting orders beside the grid — spiral, cylinder (ring), trajectory-from-file, the known-frame map of a
result folder, row- / column-major orders — against the REFERENCE's own methods (tests/golden/pose_sets.npz, written by
gen_golden.py `poses` from /root/reference/sgam/inference_pipeline.py:144-155, 206-431, 477-531 run on a bare instance)."""
import os

import numpy as np
import pytest

from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, intrinsics


def bare(data, **kw):
    s = InfiniteSceneGeneration.__new__(InfiniteSceneGeneration)
    s.data, s.step_size_denom, s.K = data, 2, intrinsics(data)
    s.anchor_poses, s.grid_transform_path, s.trajectory_shape = {}, None, "grid"
    for k, v in kw.items():
        setattr(s, k, v)
    return s


@pytest.mark.parametrize("data,tag", [("google_earth", "ge"), ("clevr-infinite", "clevr")])
@pytest.mark.parametrize("shape", ["spiral", "cylinder"])
def test_spiral_and_cylinder_match_the_reference(golden, data, tag, shape):
    g = golden("pose_sets.npz")
    s = bare(data)
    if shape == "spiral":
        s.prepare_spiral((9, 1), {})
    else:
        s.prepare_ring((9, 1), {}, horizontal_offset=0.002)
    nodes = [row[0] for row in s.transform_grid]
    assert len(nodes) == 9 and all(len(row) == 1 for row in s.transform_grid)      # steppable: pose i is transform_grid[i][0]
    assert [n["grid_coord"] for n in nodes] == [(i, 0) for i in range(9)]
    # float64 host formulas: the same operations in the same order as the reference -> bit-equal
    assert np.array_equal(np.stack([n["R"] for n in nodes]), g[f"{tag}_{shape}_R"])
    assert np.array_equal(np.stack([n["t"] for n in nodes]), g[f"{tag}_{shape}_t"])
    assert np.array_equal(np.stack([n["position"] for n in nodes]), g[f"{tag}_{shape}_position"])
    assert not any(n["visited"] for n in nodes)


def test_known_map_and_trajectory_match_the_reference(golden, tmp_path):
    g = golden("pose_sets.npz")
    np.savetxt(tmp_path / "cam0_to_world.txt", g["poses_txt"])
    for idx, i, j in g["known_files"]:
        np.save(tmp_path / f"dm_{idx:05d}_{i:02d}_{j:02d}.npy", np.zeros((2, 2), np.float32))
    s = bare("google_earth", grid_transform_path=tmp_path, trajectory_shape="trajectory", num_src=3, curr=5)
    km = s.get_known_map()
    assert np.array_equal(np.array(sorted(km.keys())), g["known_keys"])
    assert [km[k]["orig_frame_idx"] for k in sorted(km)] == list(g["known_orig_idx"])
    assert all(os.path.basename(v["rgb_path"]).startswith("im_") and v["rgb_path"].endswith(".png") for v in km.values())
    order = s.prepare_trajectory(7, km, pose_path=tmp_path / "cam0_to_world.txt")
    assert np.array_equal(np.array(order), g["traj_order"])
    nodes = [row[0] for row in s.transform_grid]
    assert np.array_equal(np.stack([n["R"] for n in nodes]), g["traj_R"])
    assert np.array_equal(np.stack([n["t"] for n in nodes]), g["traj_t"])
    assert np.array_equal(np.stack([n["position"] for n in nodes]), g["traj_position"])
    assert np.array_equal(np.array([n["visited"] for n in nodes]), g["traj_visited"])
    assert np.array_equal(np.array(sorted(s.anchor_poses.keys())), g["traj_anchor_keys"])
    assert np.array_equal(np.array(s.get_src_grid_coords((5, 0))[0]), g["traj_srcs_of_5"])
    assert tuple(s.get_closest_anchor(nodes[6])["grid_coord"]) == tuple(g["traj_closest_anchor_of_6"])
    # a trajectory that runs off the end of the pose file is refused like the reference's assert
    with pytest.raises(AssertionError):
        s.prepare_trajectory(20, km, pose_path=tmp_path / "cam0_to_world.txt")


def test_no_folder_means_nothing_known():
    assert bare("google_earth").get_known_map() == {}


def test_visiting_orders_match_the_reference(golden):
    g = golden("pose_sets.npz")
    for name in ("row_major", "column_major", "zig_zag"):
        s = bare("google_earth", output_dim=(3, 4))
        s.transform_grid = [[{"visited": False} for _ in range(4)] for _ in range(3)]
        order = getattr(s, name + "_order")()
        assert np.array_equal(np.array(order), g[name + "_3x4"]), name
        assert s.transform_grid[order[0][0]][order[0][1]]["visited"]


def test_unknown_shape_is_refused():
    with pytest.raises(NotImplementedError):
        InfiniteSceneGeneration(None, "google_earth", trajectory_shape="helix")
