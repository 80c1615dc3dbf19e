from torchrl_amd.objectives.value import functional  # noqa: F401
