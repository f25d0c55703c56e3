"""MI355X-native conditional-likelihood engine for MrBayes."""
