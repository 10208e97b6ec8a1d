#!/usr/bin/env python3
"""time_variants with a forced fused grid (SARPRO_HIP_FUSED_GRID)"""
