"""trajopt.ilqr.objects (/root/reference/trajopt/ilqr/objects.py): the containers, same names and shapes."""
from trajopt_amd.ilqr.objects import *  # noqa: F401,F403
from trajopt_amd.ilqr.objects import (AnalyticalLinearDynamics, AnalyticalQuadraticCost, LinearControl,  # noqa: F401
                                      QuadraticStateActionValue, QuadraticStateValue)
