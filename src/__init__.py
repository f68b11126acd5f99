"""The reference's ``src`` import paths, served by the MI355X-native implementation (diffusion-nlc_amd/)."""
