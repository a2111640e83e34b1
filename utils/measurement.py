"""Drop-in for the reference's utils/measurement.py (create_score_mat, PRfunc, PR_func)."""
from tf2_yolo_amd.measurement import PR_func, PRfunc, create_score_mat  # noqa: F401
