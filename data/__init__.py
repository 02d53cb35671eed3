"""Stand-in for the reference's ``data`` package: only ``data.utils.utils`` (star-imported by
main_scene_generation.py:6) is on the inference path; the training datasets are out of scope."""
