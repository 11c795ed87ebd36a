"""Import-path alias of the reference package layout (drop-in boundary); implementation: dose_prediction_amd."""
