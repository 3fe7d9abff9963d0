"""Name-holder so `from isaacgym import gymapi, gymtorch` resolves (generator-side only)."""
