"""Trajectory metrics used by the training loop and the pose evaluation (reference utils_poses/ + the ATE/ alignment it calls):
ATE / RPE of the learned camera poses against ground truth after a sim(3) alignment.  Host-side numpy, a few dozen poses."""
