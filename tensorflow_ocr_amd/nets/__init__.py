"""Host-side mirrors of the reference's `nets/` modules (same function names, argument order
and output-tensor names) running on the MI355X kernels."""
