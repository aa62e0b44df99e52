"""Host runtime of the MI355X segmentation forward pass.

`graph` / `layers` give the define-then-run surface the model code is written
against (the role TensorFlow + slim play in the reference), `plan` lowers a
graph to calls into libfss_gfx950.so, `_lib` is the ctypes binding.
"""
