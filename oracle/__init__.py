"""CPU oracle for the ICNet / PSPNet forward pass and the mIoU metric.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
package: it exists so that tests, the smoke entry point and the CPU-baseline
leg of the benchmark can check the HIP path against an independent numpy
restatement of the reference algorithm.

PARITY UNPINNED: the reference (oandrienko/fast-semantic-segmentation) ships
no tests, golden vectors or fixtures for this path, and its arithmetic lives
in an absent third-party dependency (tensorflow==1.8.0, `tf.contrib.slim`,
pinned by the reference's docs/installation.md:7).  The oracle therefore
restates TensorFlow 1.8's published op semantics and is anchored on

  * the reference's own call sites (cited per function as file:line),
  * the three published parameter counts (docs/model_zoo.md:26-28), which
    `oracle.nets` reproduces exactly, and
  * an independent second implementation of every op (torch CPU), checked in
    tests/test_oracle_ops.py.
"""
