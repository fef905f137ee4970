"""Constants shared by tests/gen_golden.py (which needs /root/reference) and the parity tests (which must not import it)."""
FULL_LAMBDAS = {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1, "lambda_sparsity": 0.01}
