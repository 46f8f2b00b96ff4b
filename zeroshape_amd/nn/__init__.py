"""Inference layers of the image encoders on the HIP library (channels-last fp32).
`pack` turns reference-layout parameters into the kernels' layouts (host, once per checkpoint);
`ops` are thin launch wrappers.  GPU only: nothing here has a CPU or PyTorch-op fallback."""
