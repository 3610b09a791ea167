"""Import-path shim: the reference's entry-point locations (`cosmos_predict1/diffusion/inference/gen3c_*.py`) resolve, inside this
repository, to the MI355X-native implementations under `gen3c_amd/`. Nothing is implemented here - see SURVEY.md 8b "CLI" seam."""
