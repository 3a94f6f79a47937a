"""trajopt_amd - MI355X-native batched iLQR / Riccati engine behind the trajopt.ilqr / trajopt.riccati API."""
__version__ = "0.1.0"
