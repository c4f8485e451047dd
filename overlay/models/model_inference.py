"""Drop-in for the reference's models/model_inference.py."""
from dino_tracker_amd.model_inference import (ModelInference, generate_trajectories, generate_trajectory,  # noqa: F401
                                              generate_trajectory_input)
