"""Task constants of the simulated ALOHA tasks (values from reference constants_org.py:10-64; the fork's
constants.py no longer defines them)."""

DT = 0.02          # constants_org.py:63
FPS = 50           # constants_org.py:64
SIM_CAMERAS = ["top", "left_wrist", "right_wrist"]

SIM_TASK_CONFIGS = {
    "sim_transfer_cube_scripted": {"dataset_dir": "data/sim_transfer_cube_scripted", "num_episodes": 50,
                                   "episode_len": 400, "camera_names": list(SIM_CAMERAS)},
    "sim_transfer_cube_human": {"dataset_dir": "data/sim_transfer_cube_human", "num_episodes": 50,
                                "episode_len": 400, "camera_names": ["top"]},
    "sim_insertion_scripted": {"dataset_dir": "data/sim_insertion_scripted", "num_episodes": 50,
                               "episode_len": 400, "camera_names": list(SIM_CAMERAS)},
    "sim_insertion_human": {"dataset_dir": "data/sim_insertion_human", "num_episodes": 50,
                            "episode_len": 500, "camera_names": ["top"]},
    # the BASELINE.json metric configuration: 4 cameras
    "sim_synthetic_4cam": {"dataset_dir": None, "num_episodes": 50, "episode_len": 400,
                           "camera_names": ["top", "left_wrist", "right_wrist", "angle"]},
}
