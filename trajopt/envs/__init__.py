"""trajopt.envs - the env classes of /root/reference/trajopt/envs/__init__.py that the hot path covers, gym-free."""
from trajopt_amd.envs import *  # noqa: F401,F403
from trajopt_amd.envs import make  # noqa: F401
