# import-only stand-in (graphics_pgl); never used on the step path
