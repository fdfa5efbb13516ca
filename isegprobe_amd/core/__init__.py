"""Host-side mirror of the reference's ``core`` package for the dense-feature path.

Same module paths, class names, constructor arguments and registries as
havrylovv/iSegProbe's ``core.model`` / ``core.utils.model_builder`` / ``core.inference``,
with every tensor op executed by the HIP library (isegprobe_amd/csrc).  Call
``isegprobe_amd.install_as_core()`` to alias this package as top-level ``core`` so that
reference checkpoints (which pickle ``core.utils.model_builder.ModelBuilder`` and resolve
``core.model.iseg_probe_model.iSegProbeModel`` by dotted path) load unchanged.
"""
